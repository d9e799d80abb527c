"""CPU: the C-ABI library loads and exports every function include/*.h declares.
No compute calls here (there is no GPU on this box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [
    "include/coolmic_hip.h",
    "include/coolmic-dsp/ro-compat.h",
    "include/coolmic-dsp/coolmic-dsp.h",
    "include/coolmic-dsp/logging.h",
    "include/coolmic-dsp/iohandle.h",
    "include/coolmic-dsp/transform.h",
    "include/coolmic-dsp/vumeter.h",
    "include/coolmic-dsp/snddev.h",
    "include/coolmic-dsp/tee.h",
    "include/coolmic-dsp/util.h",
    "include/coolmic-dsp/group.h",
]
DECL = re.compile(r"\b((?:cmhip|coolmic)_[a-z0-9_]+)\s*\(")


def _declared(path):
    text = open(os.path.join(ROOT, path)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)           # comments
    text = re.sub(r"^\s*#\s*define[^\n]*(\\\n[^\n]*)*", "", text, flags=re.M)  # macros
    names = set()
    for m in DECL.finditer(text):
        n = m.group(1)
        if n.endswith("_t"):
            continue
        names.add(n)
    return names


def test_headers_compile_as_c_and_cxx(tmp_path):
    import subprocess
    src = "\n".join('#include <%s>' % h.split("include/", 1)[1] for h in HEADERS) + "\nint main(void){return 0;}\n"
    for comp, ext, std in (("gcc", "c", "-std=gnu11"), ("g++", "cpp", "-std=c++17")):
        f = tmp_path / ("t." + ext)
        f.write_text(src)
        subprocess.run([comp, std, "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(f)], check=True)


def test_every_declared_symbol_is_exported(cm):
    declared = set()
    for h in HEADERS:
        declared |= _declared(h)
    assert len(declared) > 50
    missing = [n for n in sorted(declared) if not hasattr(cm.lib, n)]
    assert not missing, missing
    # and the Python mirror knows each of them
    unbound = [n for n in sorted(declared) if n not in cm.SIGNATURES]
    assert not unbound, unbound


def test_result_struct_layout_matches_reference(cm):
    """coolmic_vumeter_result_t is 192 bytes on LP64 with the offsets SURVEY 8(a) a9 lists"""
    import ctypes as C
    R = cm.VuResult
    assert C.sizeof(R) == 192
    offs = {n: getattr(R, n).offset for n, _ in R._fields_}
    assert offs == {"rate": 0, "channels": 4, "frames": 8, "global_peak": 16, "global_power": 24,
                    "channel_peak": 32, "channel_power": 64}


def test_library_reports_itself(cm):
    assert b"gfx950" in cm.lib.cmhip_version()
    assert cm.lib.coolmic_feature_check(b"accel:hip/gfx950") == 1
    assert cm.lib.coolmic_feature_check(b"driver:sine") == 1
    assert cm.lib.coolmic_feature_check(b"driver:si") == 0
    assert cm.lib.coolmic_feature_check(b"encode:ogg/vorbis") == 0
    assert cm.lib.coolmic_feature_check(None) == cm.ERROR_FAULT
    assert cm.lib.coolmic_feature_check(b"") == cm.ERROR_INVAL
    assert cm.lib.coolmic_error2string(-10) == b"Invalid argument"
    assert cm.lib.coolmic_error2string(-9) == b"Bad address"
    assert cm.lib.coolmic_error2string(12345) == b"(unknown)"


def test_no_oracle_in_the_product():
    """the product must not reach into oracle/ (nor carry a CPU fallback)"""
    pkg = os.path.join(ROOT, "libcoolmic-dsp_amd")
    for base, _dirs, files in os.walk(pkg):
        if os.sep + "build" in base or os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "sine_table.c" or f == "__init__.py", \
                    os.path.join(base, f)
    import subprocess
    so = os.path.join(pkg, "lib", "libcoolmic-dsp-hip.so")
    deps = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in deps


REFERENCE = "/root/reference"
# the translation units the library replaces inside the reference's own build and the
# reference headers that declare what those units define
REPLACED_UNITS = {"transform.c": "transform.h", "vumeter.c": "vumeter.h", "iohandle.c": "iohandle.h",
                  "tee.c": "tee.h", "logging.c": "logging.h", "coolmic-dsp.c": "coolmic-dsp.h"}


def _exported(so):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    return {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TWB"}


def test_dropin_build_exports_what_the_replaced_units_define():
    """The recipe of INTEGRATION.md 3 takes six translation units out of the reference's build.  Every function
    the reference's headers declare for those units must come out of `make dropin`
    (lib/libcoolmic-dsp-hip-dropin.so); what stays the reference's -- the sources snddev*.c with their driver
    vtable, util.c -- must NOT be defined by it; and the recipe's filter-out list is exactly that set."""
    import subprocess
    pkg = os.path.join(ROOT, "libcoolmic-dsp_amd")
    subprocess.run(["make", "-s", "-C", pkg, "dropin"], check=True)
    names = _exported(os.path.join(pkg, "lib", "libcoolmic-dsp-hip-dropin.so"))
    recipe = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"filter-out ([^,]*),", recipe)
    assert m, "INTEGRATION.md has no filter-out line"
    assert set(m.group(1).split()) == set(REPLACED_UNITS), m.group(1)
    kept = {"coolmic_snddev_new", "coolmic_snddev_get_iohandle", "coolmic_snddev_attach_iohandle",
            "coolmic_snddev_iter", "coolmic_util_ahsv2argb", "coolmic_util_power2hue", "coolmic_util_peak2hue"}
    assert not (names & kept), names & kept
    # internal glue between the units is not part of the ABI
    assert not [n for n in names if n.startswith(("coolmic_transform_records", "coolmic_tee_reader", "cmhip_vu_raw"))]
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference headers are not on this machine; the export list was checked without them")
    declared = set()
    for unit, header in REPLACED_UNITS.items():
        assert os.path.exists(os.path.join(REFERENCE, "src", unit)), unit
        text = open(os.path.join(REFERENCE, "include", "coolmic-dsp", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^\s*#\s*define[^\n]*(\\\n[^\n]*)*", "", text, flags=re.M)
        declared |= {n for n in re.findall(r"\b(coolmic_[a-z0-9_]+)\s*\(", text) if not n.endswith("_t")}
    assert len(declared) >= 21, sorted(declared)
    missing = sorted(declared - names)
    assert not missing, missing
    # the stand-alone library (with its own sources and helpers) exports them too
    full = _exported(os.path.join(pkg, "lib", "libcoolmic-dsp-hip.so"))
    assert not sorted(declared - full)


def _igloo_uses(text):
    """igloo identifiers of a C text -> {name: set of argument counts} (None: used without a call).  Comments
    and string literals are dropped first; arguments are counted at the top nesting level."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r'"(\\.|[^"\\])*"', '""', text)
    uses = {}
    for m in re.finditer(r"\bigloo_\w+", text):
        name, i = m.group(0), m.end()
        while i < len(text) and text[i] in " \t\\\n":
            i += 1
        arity = None
        if i < len(text) and text[i] == "(":
            depth, args, seen, j = 0, 0, False, i
            while j < len(text):
                c = text[j]
                if c in "([{":
                    depth += 1
                elif c in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                elif c == "," and depth == 1:
                    args += 1
                elif depth == 1 and not c.isspace():
                    seen = True
                j += 1
            arity = args + 1 if seen else 0
        uses.setdefault(name, set()).add(arity)
    return uses


def test_igloo_glue_uses_the_references_igloo_surface():
    """`make dropin IGLOO=1` turns the stages into libigloo objects through csrc/ro_igloo.h.
    libigloo is not in this image, so the file cannot be compiled here; what can be held to account is that
    every igloo_* identifier it uses is one the reference's own sources use, called with the same number of
    arguments (ref: src/transform.c:54-62,72; src/vumeter.c:59-67,76; src/iohandle.c:41-52,62; src/tee.c:71-81,227)
    -- and that the default build never sees the file."""
    csrc = os.path.join(ROOT, "libcoolmic-dsp_amd", "csrc")
    glue = open(os.path.join(csrc, "ro_igloo.h")).read()
    own = _igloo_uses(glue)
    assert {"igloo_RO_PUBLIC_TYPE", "igloo_RO_TYPEDECL_FREE", "igloo_RO_TO_TYPE", "igloo_ro_new_raw", "igloo_ro_ref",
            "igloo_ro_unref", "igloo_ro_t"} <= set(own), sorted(own)
    # out of the default build: ro_glue.h takes it under the switch only, it refuses to compile without it, and
    # the Makefile leaves csrc/ro.c out exactly when IGLOO is set
    assert re.search(r"#ifdef COOLMIC_DSP_USE_LIBIGLOO\s*\n#include \"ro_igloo.h\"\s*\n#else", open(os.path.join(csrc, "ro_glue.h")).read())
    assert "#error" in glue
    mk = open(os.path.join(ROOT, "libcoolmic-dsp_amd", "Makefile")).read()
    assert re.search(r"ifdef IGLOO\n(?:.*\n)*?DROPIN_C\s*:=\s*\$\(filter-out ro\.c,\$\(DROPIN_C\)\)", mk)
    assert "-DCOOLMIC_DSP_USE_LIBIGLOO" in mk and "COOLMIC_DSP_USE_LIBIGLOO" not in mk.split("ifdef IGLOO")[0]
    # the stages themselves speak only the glue's vocabulary: no libigloo name of their own beyond the two
    # the public ro-compat.h supplies in both builds
    for unit in ("iohandle.c", "transform.c", "vumeter.c", "tee.c"):
        names = set(_igloo_uses(open(os.path.join(csrc, unit)).read()))
        assert names <= {"igloo_ro_t", "igloo_ro_base_t", "igloo_RO_NULL"}, (unit, names)
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        pytest.skip("the reference's sources are not on this machine; the glue was checked without them")
    ref = {}
    for d in ("src", os.path.join("include", "coolmic-dsp")):
        for f in sorted(os.listdir(os.path.join(REFERENCE, d))):
            if f.endswith((".c", ".h")):
                for name, ar in _igloo_uses(open(os.path.join(REFERENCE, d, f), errors="replace").read()).items():
                    ref.setdefault(name, set()).update(ar)
    unknown = sorted(set(own) - set(ref))
    assert not unknown, "igloo names the reference never uses: %s" % unknown
    wrong = {n: (sorted(a, key=str), sorted(ref[n], key=str)) for n, a in own.items() if not a <= ref[n]}
    assert not wrong, "argument counts differ from the reference's usage: %s" % wrong
    # and the one private header it includes is the reference's own
    assert os.path.exists(os.path.join(REFERENCE, "src", "types_private.h"))


def test_no_scalar_load_with_a_register_and_an_immediate_offset():
    """On gfx950 with ROCm 7.2 an `s_load_dword sdst, sbase, soffset offset:imm` came back from sbase + imm alone:
    the register offset did not reach the address (round 4, profiles/NOTES_r04.md -- a uniform, run-time indexed
    load of a stream's parameters in the stereo read-only kernel; golden vector G4 caught it on the GPU).  The
    kernels avoid the form (fixed offsets + scalar selects); this holds the generated assembly of all three kernel
    files to that, on the CPU, before anything reaches a GPU.  It also keeps scratch memory out of the hot kernels:
    a spilled register means a private segment per wave."""
    import subprocess
    pkg = os.path.join(ROOT, "libcoolmic-dsp_amd")
    subprocess.run(["make", "-s", "-C", pkg, "asm"], check=True)
    bad = []
    for name in ("k_block", "k_eq", "k_misc"):
        text = open(os.path.join(pkg, "build", name + ".s")).read()
        assert "s_load_dword" in text and ".amdhsa_kernel" in text, name
        for ln in text.splitlines():
            if re.search(r"^\s*s_(buffer_)?load_dword\w*\s+\S+,\s*s\[\d+:\d+\],\s*s\d+\s+offset:", ln):
                bad.append((name, ln.strip()))
    assert not bad, bad[:5]
    usage = open(os.path.join(pkg, "build", "k_block.usage.txt")).read()
    spills = {}
    for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", usage, flags=re.S):
        spills[m.group(1)] = int(m.group(2))
    hot = [n for n in spills if "k_run_fast" in n and ("Li16E" in n or "ELb1ELb0ELb1ELi4ELi4E" in n)]
    assert hot, "the mono / stereo kernels of configs 2, 4, 5 and of the read-only leg were not found"
    assert not [n for n in hot if spills[n]], [n for n in hot if spills[n]]


def test_host_feature_tokens_of_the_dropin_build():
    """inside the reference's build coolmic_features() lists the HOST's encoders and drivers (handed in by its
    Makefile) plus the token of this path (ref: src/coolmic-dsp.c:64-83)"""
    import ctypes as C
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "libcoolmic-dsp_amd"), "dropin"], check=True)
    lib = C.CDLL(os.path.join(ROOT, "libcoolmic-dsp_amd", "lib", "libcoolmic-dsp-hip-dropin.so"))
    lib.coolmic_features.restype = C.c_char_p
    toks = lib.coolmic_features().split(b" ")
    assert toks[0] == b"features" and b"accel:hip/gfx950" in toks and b"encode:ogg/vorbis" in toks
    assert lib.coolmic_feature_check(b"encode:ogg/opus") == 1 and lib.coolmic_feature_check(b"driver:sine") == 0


def _macros(path):
    """object-like macros of a header: name -> replacement text (comments and spacing stripped)"""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+([A-Za-z_][A-Za-z_0-9]*)(?![A-Za-z_0-9(])[ \t]*(.*)$", text, flags=re.M):
        out[m.group(1)] = " ".join(m.group(2).split())
    return out


def test_public_macros_equal_the_references():
    """A host's sources compile against these headers instead of the reference's: every object-like macro the
    reference's versions of the shipped headers define (error numbers, feature tokens, driver names, limits,
    log levels ...) is defined here with the same replacement text.  (Include guards aside; function-like
    macros are compared by test_headers_compile_as_c_and_cxx's clients.)"""
    if not os.path.isdir(os.path.join(REFERENCE, "include", "coolmic-dsp")):
        pytest.skip("the reference headers are not on this machine")
    missing, different = [], []
    for h in ("coolmic-dsp.h", "iohandle.h", "transform.h", "vumeter.h", "tee.h", "snddev.h", "logging.h", "util.h"):
        ref = _macros(os.path.join(REFERENCE, "include", "coolmic-dsp", h))
        own = _macros(os.path.join(ROOT, "include", "coolmic-dsp", h))
        for name, value in ref.items():
            if name.startswith("__COOLMIC_DSP_") and value == "":
                continue                                 # include guard
            if name not in own:
                missing.append((h, name))
            elif own[name] != value:
                different.append((h, name, value, own[name]))
    assert not missing, missing
    assert not different, different


def _prototypes(path):
    """coolmic_* function declarations of a header: name -> (return type, parameter list), spacing normalised,
    attributes dropped"""
    import re
    t = open(path).read()
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//.*", " ", t)
    t = re.sub(r"^\s*#.*$", " ", t, flags=re.M)
    t = re.sub(r"__attribute__\s*\(\((?:[^()]|\([^()]*\))*\)\)", " ", t)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*\n\t]*?)\b(coolmic_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", t):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret:
            continue
        tight = lambda x: re.sub(r"\s*([\*\(\),])\s*", r"\1", x)
        out[name] = (tight(ret), tight(args))
    return out


def test_prototypes_equal_the_references():
    """every function the reference's versions of the shipped headers declare is declared here with the same
    return type and the same parameter list, token for token (ref: include/coolmic-dsp/*.h)"""
    if not os.path.isdir(os.path.join(REFERENCE, "include", "coolmic-dsp")):
        pytest.skip("the reference headers are not on this machine")
    seen = 0
    for h in ("coolmic-dsp.h", "iohandle.h", "transform.h", "vumeter.h", "tee.h", "snddev.h", "logging.h", "util.h"):
        ref = _prototypes(os.path.join(REFERENCE, "include", "coolmic-dsp", h))
        own = _prototypes(os.path.join(ROOT, "include", "coolmic-dsp", h))
        for name, proto in ref.items():
            assert name in own, (h, name)
            assert own[name] == proto, (h, name, proto, own[name])
            seen += 1
    assert seen >= 28                                     # (the eight headers declare 28 functions)
