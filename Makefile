# convenience targets; the authoritative entry points are __graft_entry__.build()/smoke(), pytest and bench.py
.PHONY: build test gpu-test smoke bench profiles clean test-asan
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
# AddressSanitizer + UndefinedBehaviorSanitizer on the CPU side (sanitizers cannot run on the GPU boxes of this pool):
# the oracle, the pybind11 host module and the fake RCCL transport are rebuilt with -fsanitize=address,undefined under build/asan/ and
# the CPU test files that drive them run against those builds (the interpreter itself is not instrumented: leak detection off).
ASAN_FLAGS := -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -fPIC
PYINC := $(shell python3 -c "import sysconfig,pybind11;print('-I'+sysconfig.get_paths()['include'],'-I'+pybind11.get_include())")
PYEXT := $(shell python3 -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
test-asan: build
	mkdir -p build/asan/neutfem
	gcc $(ASAN_FLAGS) -std=c99 -Wall -Wextra -shared -o build/asan/libnf_oracle.so oracle/nf_oracle.c -lm
	g++ $(ASAN_FLAGS) -std=c++17 -shared -fvisibility=hidden $(PYINC) -Iinclude -o build/asan/neutfem/_neutfem_eigen$(PYEXT) neutfem_amd/csrc/host_module.cpp \
	    -Lneutfem_amd/lib -lneutfem_hip -Wl,-rpath,$(CURDIR)/neutfem_amd/lib
	g++ $(ASAN_FLAGS) -std=c++17 -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o build/asan/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt
	LD_PRELOAD="$$(gcc -print-file-name=libasan.so) $$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	    NF_ORACLE_LIB=$(CURDIR)/build/asan/libnf_oracle.so NEUTFEM_MODULE_DIR=$(CURDIR)/build/asan NEUTFEM_ASAN_FAKE_RCCL=$(CURDIR)/build/asan/libfake_rccl.so \
	    python -m pytest tests/test_oracle.py tests/test_boundary.py tests/test_asan_build.py -x -q -m "not gpu" -p no:cacheprovider -k "not reproduces_golden and not cache_is_current"   # those two pin the bits of the -O3 build
clean:
	$(MAKE) -C neutfem_amd/csrc clean || true
	rm -f oracle/*.so tests/fake_rccl/*.so
