# Convenience targets; the real build rules live in loltracer_amd/csrc/Makefile and oracle/Makefile.
.PHONY: all build test test-gpu bench clean
all: build
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	$(MAKE) -C loltracer_amd/csrc clean
	$(MAKE) -C oracle clean
