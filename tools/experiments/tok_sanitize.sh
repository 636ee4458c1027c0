#!/bin/bash
# The host tokenizer (csrc/tokenize.hip, no device code) under ThreadSanitizer and AddressSanitizer + UBSan, CPU only:
#   tok_tsan_harness.cpp  three callers at once (the worker pool, and the spawn path a busy pool falls back to)
#   tok_asan_harness.cpp  texts of every unit size in heap blocks of their exact size (an over-read of the 16-byte window scan
#                         or of the 8-byte key loads would trip), vocabulary keys of 1 - 40 bytes, 1 - 3 threads
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"
tmp="$(mktemp -d)"
echo 'int tt_fail(int c, const char *, ...) { return c; }' > "$tmp/stub.cpp"
inc="-I$root/twotowermlretrieval_amd/csrc -I$root/include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__"
g++ -O1 -g -fsanitize=thread -std=c++17 -w -x c++ $inc -c "$root/twotowermlretrieval_amd/csrc/tokenize.hip" -o "$tmp/t.o"
g++ -O1 -g -fsanitize=thread "$root/tools/experiments/tok_tsan_harness.cpp" "$tmp/stub.cpp" "$tmp/t.o" -o "$tmp/tsan" -pthread -ldl
"$tmp/tsan"
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -w -x c++ $inc -c "$root/twotowermlretrieval_amd/csrc/tokenize.hip" -o "$tmp/a.o"
g++ -O1 -g -fsanitize=address,undefined "$root/tools/experiments/tok_asan_harness.cpp" "$tmp/stub.cpp" "$tmp/a.o" -o "$tmp/asan" -pthread -ldl
"$tmp/asan"
rm -rf "$tmp"
