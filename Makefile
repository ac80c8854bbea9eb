# Build recipe outside Python (the same commands __graft_entry__.build() runs).
#   make            libserhip.so for gfx950 + the plain-C example host
#   make fuzz       host-only sanitizer builds of the two parsers that read untrusted bytes (scripts/fuzz_*.cpp)
HIPCC ?= /opt/rocm/bin/hipcc
ROCM  ?= /opt/rocm
CSRC  := ser_amd/csrc
LIB   := ser_amd/libserhip.so

all: $(LIB) examples/host_c_abi

$(LIB): $(CSRC)/serhip.hip $(wildcard $(CSRC)/*.hpp) include/serhip.h
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -shared -fPIC -o $@ $(CSRC)/serhip.hip

examples/host_c_abi: examples/host_c_abi.c include/serhip.h $(LIB)
	gcc -O2 -Wall -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -Iinclude $< -o $@ -Lser_amd -l:libserhip.so \
	    -L$(ROCM)/lib -lamdhip64 '-Wl,-rpath,$$ORIGIN/../ser_amd' -Wl,-rpath,$(ROCM)/lib

fuzz: /tmp/fuzz_flac /tmp/fuzz_wav
/tmp/fuzz_%: scripts/fuzz_%.cpp $(wildcard $(CSRC)/*.hpp)
	g++ -std=c++17 -O1 -g -pthread -fsanitize=address,undefined -fno-sanitize-recover=all -I. $< -o $@

clean:
	rm -f $(LIB) examples/host_c_abi

.PHONY: all fuzz clean
