# Build of the MI355X k-mer counting engine (gfx950 only) and its test infrastructure.
#   make            -> yak_amd/libyak_amd.so (HIP kernels + C ABI), yak_amd/yak-amd (CLI),
#                      tools/ (synthetic reads), oracle/ (CPU checker; + oracle/_ref if the
#                      reference sources are present)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-result -Wno-unused-value
CSRC    = yak_amd/csrc

all: lib cli tools oracle

lib: yak_amd/libyak_amd.so
cli: yak_amd/yak-amd

yak_amd/kernels.o: $(CSRC)/kernels.hip $(wildcard $(CSRC)/kern_*.inc) $(CSRC)/yk_device.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
yak_amd/engine.o: $(CSRC)/engine.cpp $(CSRC)/engine_int.h $(CSRC)/engine.h $(CSRC)/yk_device.h include/yak.h include/yak_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
yak_amd/layout.o: $(CSRC)/layout.cpp $(CSRC)/engine_int.h $(CSRC)/engine.h $(CSRC)/yk_device.h include/yak.h include/yak_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
yak_amd/pool.o: $(CSRC)/pool.cpp $(CSRC)/engine.h $(CSRC)/yk_device.h include/yak.h include/yak_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
HOSTDEPS = $(CSRC)/yak_host.h $(CSRC)/pgz.h $(CSRC)/engine.h $(CSRC)/yk_device.h include/yak.h include/yak_amd.h
yak_amd/yak_api.o: $(CSRC)/yak_api.cpp $(HOSTDEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
yak_amd/yak_reader.o: $(CSRC)/yak_reader.cpp $(HOSTDEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
yak_amd/yak_multi.o: $(CSRC)/yak_multi.cpp $(HOSTDEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
yak_amd/libyak_amd.so: yak_amd/kernels.o yak_amd/pool.o yak_amd/engine.o yak_amd/layout.o yak_amd/yak_api.o yak_amd/yak_reader.o yak_amd/yak_multi.o $(CSRC)/libyak_amd.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -Wl,-Bsymbolic -Wl,--version-script=$(CSRC)/libyak_amd.map -o $@ $(filter %.o,$^) -lz

yak_amd/yak-amd: $(CSRC)/main.c include/yak.h yak_amd/libyak_amd.so
	gcc -O2 -Wall -Iinclude $(CSRC)/main.c -o $@ -Lyak_amd -lyak_amd -Wl,-rpath,'$$ORIGIN' -lz

tools:
	$(MAKE) -C tools
oracle: lib
	$(MAKE) -C oracle all ref

clean:
	rm -f yak_amd/*.o yak_amd/libyak_amd.so yak_amd/yak-amd
	$(MAKE) -C tools clean
	$(MAKE) -C oracle clean
.PHONY: all lib cli tools oracle clean
