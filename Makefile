# Convenience targets; the real build rules live in towr_amd/csrc/Makefile and oracle/Makefile.
.PHONY: all test test-gpu example clean
all:
	$(MAKE) -C towr_amd/csrc
	$(MAKE) -C oracle
test: all
	python -m pytest tests -q -m "not gpu"
test-gpu: all
	python -m pytest tests -q -m gpu
example: all
	mkdir -p examples/_build
	/opt/rocm/bin/hipcc -std=c++17 -O2 -Wall -I include examples/sweep_example.cc -L towr_amd -ltowr_amd \
	  -Wl,-rpath,$(CURDIR)/towr_amd -o examples/_build/sweep_example
clean:
	$(MAKE) -C towr_amd/csrc clean
	rm -rf oracle/_build examples/_build
