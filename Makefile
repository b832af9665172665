# Convenience targets; the driver's contract is __graft_entry__.build()/smoke() and bench.py.
.PHONY: build test test-gpu bench smoke example clean

build:                      ## hipcc --offload-arch=gfx950 -> phyx_amd/libphyx_amd.so; gcc -> oracle/liboracle.so (+ oracle/_ref shims if /root/reference exists)
	python -c "import __graft_entry__ as g; g.build()"

test: build                 ## CPU suite (no GPU needed)
	python -m pytest tests -q -m "not gpu"

test-gpu: build             ## parity suite on an MI355X
	python -m pytest tests -q -m gpu

smoke: build
	python __graft_entry__.py --smoke

bench: build
	python bench.py

example: build              ## the C ABI from plain C
	gcc -std=c11 -O2 -Wall -Iinclude examples/drop_in.c -Lphyx_amd -lphyx_amd -Wl,-rpath,$(CURDIR)/phyx_amd -o examples/drop_in

clean:
	rm -f phyx_amd/libphyx_amd.so oracle/liboracle.so examples/drop_in
	rm -rf oracle/_ref
