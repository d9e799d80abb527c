# Convenience targets; the real build files are libcoolmic-dsp_amd/Makefile (the library) and oracle/Makefile
# (the CPU checker, test infrastructure).
#
#   make            library + oracle (+ oracle/_ref where the reference's sources are present)
#   make dropin     the library as it goes into the reference's own build
#   make test-cpu   what runs without a GPU:  python -m pytest tests -m "not gpu"
#   make test-gpu   parity tests proper (needs an MI355X):  python -m pytest tests -m gpu
#   make bench      python bench.py (one JSON line; N = 1)

REF ?= /root/reference

all:
	$(MAKE) -C libcoolmic-dsp_amd
	$(MAKE) -C oracle
	@if [ -f $(REF)/src/util.c ]; then $(MAKE) -C oracle _ref REF=$(REF); fi

dropin: all
	$(MAKE) -C libcoolmic-dsp_amd dropin

test-cpu: all
	python -m pytest tests -q -m "not gpu"

test-gpu: all
	python -m pytest tests -q -m gpu

bench: all
	python bench.py

clean:
	rm -rf libcoolmic-dsp_amd/build libcoolmic-dsp_amd/lib
	$(MAKE) -C oracle clean

.PHONY: all dropin test-cpu test-gpu bench clean
