#!/bin/bash
# Host tokenizer variants timed on THIS machine's cores (no GPU, no Python): usage tok_harness.sh [threads ...]
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"
tmp="$(mktemp -d)"
echo 'int tt_fail(int c, const char *, ...) { return c; }' > "$tmp/stub.cpp"
for defs in "" "-DTT_TOK_HUGEPAGES=0" "-DTT_TOK_LAG=32" "-DTT_TOK_NO_SIMD"; do
    g++ -O3 -std=c++17 -w -x c++ $defs -I"$root/twotowermlretrieval_amd/csrc" -I"$root/include" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
        -c "$root/twotowermlretrieval_amd/csrc/tokenize.hip" -o "$tmp/tok.o"
    g++ -O2 "$root/tools/experiments/tok_harness.cpp" "$tmp/stub.cpp" "$tmp/tok.o" -o "$tmp/h" -pthread -ldl
    echo "== variant: ${defs:-product}"
    "$tmp/h" "${@:-1}"
done
lscpu | grep -i -E "model name|L2 cache|L3 cache" || true
grep -H . /sys/kernel/mm/transparent_hugepage/enabled || true
rm -rf "$tmp"
