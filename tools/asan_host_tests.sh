#!/bin/bash
# Runs the host-only C++ (tokenizer.cpp, words.cpp, results.cpp: ASan + UBSan) and the host side of host.hip / capi.hip (ASan + UBSan)
# through the CPU tests.
# The three files are rebuilt instrumented and linked with the regular HIP objects into a scratch copy of the library, which
# replaces whisperkit_amd/libwhisperhip.so for the duration of the run.  Needs no GPU.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/whisperkit_amd/csrc; OUT=${TMPDIR:-/tmp}/wh_asan; mkdir -p $OUT
CLANG=/opt/rocm/lib/llvm/bin/clang++
make -C $CS -j8 >/dev/null
for f in tokenizer words results; do
  $CLANG -x c++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include -I$CS -c $CS/$f.cpp -o $OUT/$f.o
done
for f in host capi; do   # host side of the HIP files that carry host logic (device code is left alone)
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wno-unused-result -Wno-unused-value \
    -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -c $CS/$f.hip -o $OUT/$f.o
done
(cd $CS && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 mel.o gemm.o layernorm.o attention.o decoder.o decoder32.o xabs.o beam.o comm.o $OUT/capi.o $OUT/host.o \
  $OUT/tokenizer.o $OUT/words.o $OUT/results.o -o $OUT/libwhisperhip.so -lz -ldl -fsanitize=address,undefined -shared-libsan)
cp $ROOT/whisperkit_amd/libwhisperhip.so $OUT/libwhisperhip.orig.so
trap 'cp $OUT/libwhisperhip.orig.so $ROOT/whisperkit_amd/libwhisperhip.so' EXIT
cp $OUT/libwhisperhip.so $ROOT/whisperkit_amd/libwhisperhip.so
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $ROOT && LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_tokenizer_text.py tests/test_results_formats.py tests/test_abi_host.py tests/test_comm_abi.py -x -q -p no:cacheprovider \
  --deselect tests/test_abi_host.py::test_plain_c_host_links_and_runs_host_entry_points   # (gcc links the example against the instrumented library without the sanitizer runtimes)
