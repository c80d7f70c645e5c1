set -e
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm scripts/rowreg_bench.hip"
$B -DWALK=2 -o /tmp/rb_w2 2>/dev/null
for T in 80 120 200 400 1000; do
echo "== $T trees: 256 threads x groups of 16 | 512 threads x groups of 32 | 512 x 16"
timeout 120 /tmp/rb_w2 2 $T 1048576 2 16 1 256 | head -1
timeout 120 /tmp/rb_w2 2 $T 1048576 2 32 1 512 | head -1
timeout 120 /tmp/rb_w2 2 $T 1048576 2 16 1 512 | head -1
done
