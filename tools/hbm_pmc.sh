#!/bin/bash
# HBM bytes per launch of a bench workload from the L2's memory-side counters, collected and
# corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes; FETCH_SIZE doubled on gfx950 (128-byte requests
# of a wide coalesced stream are tallied at 64 B), WRITE_SIZE as is.  Run on the GPU box from the
# repo root:  tools/hbm_pmc.sh c3   ->  gpurun_out/pmc_c3.json (copy to profiles/ to keep it)
#             tools/hbm_pmc.sh c2ro below   ->  gpurun_out/pmc_c2ro_below.json (the gain form of the read-only runs)
set -e
WL=${1:-c2}
GAIN=${2:-general}
TAG=$WL; [ "$GAIN" = general ] || TAG=${WL}_$GAIN
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/hbm_pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    # (--no-extras: every dispatch of the kernel in the trace is a step of the bench, on arrays where hipMalloc first put them)
    rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 "$ROOT/bench.py" --workload "$WL" --gain "$GAIN" \
        --steps 5 --warmup 1 --no-cpu --no-extras > "$OUT/$c.json" 2> "$OUT/$c.err"
done
python3 - "$OUT" "$WL" "$ROOT/gpurun_out/pmc_$TAG.json" "$GAIN" <<'PY'
import csv, glob, json, sys
out, wl, dest, gain = sys.argv[1:5]
line = json.load(open(out + "/FETCH_SIZE.json"))
kern = "k_eq_pipe" if wl == "c3" else "k_run_"
raw, name = {}, None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"] and row["Counter_Name"] == c:
                vals.append(float(row["Counter_Value"]))
                name = row["Kernel_Name"]
    raw[c] = {"dispatches": len(vals), "mean_KiB": sum(vals) / len(vals), "min_KiB": min(vals), "max_KiB": max(vals)}
rd = raw["FETCH_SIZE"]["mean_KiB"] * 1024 * 2
wr = raw["WRITE_SIZE"]["mean_KiB"] * 1024
alg = line["roofline"]["algorithmic_bytes_per_launch"]
res = {"round": "round 4", "workload": wl, "gain_form": gain, "streams": line["config"]["streams_per_gpu"], "channels": line["config"]["channels"],
       "frames": line["config"]["frames_per_launch"], "kernel": name,
       "command": "tools/hbm_pmc.sh %s %s (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, one counter per pass, bench.py --gain %s --steps 5 --warmup 1 --no-cpu --no-extras)" % (wl, gain, gain),
       "raw": raw,
       "correction": "FETCH_SIZE x 1024 B x 2 (gfx950 tallies the 128-B requests of a wide coalesced stream at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE x 1024 B as is",
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg}
json.dump(res, open(dest, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("workload", "kernel", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch",
                                      "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}, indent=1))
PY
