# convenience targets; the authoritative entry points are __graft_entry__.build()/smoke(), pytest and bench.py
.PHONY: build test gpu-test smoke bench profiles clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
gpu-test: build
	python -m pytest tests -x -q -m gpu
smoke: build
	python -c "import __graft_entry__ as g; g.smoke()"
bench: build
	python bench.py
profiles: build
	bash profiles/collect.sh
clean:
	$(MAKE) -C neutfem_amd/csrc clean || true
	rm -f oracle/*.so tests/fake_rccl/*.so
