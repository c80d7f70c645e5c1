set -e
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm scripts/rowreg_bench.hip"
$B -DWALK=2 -o /tmp/rb_w2 2>/dev/null
$B -DWALK=2 -DPREFETCH -o /tmp/rb_w2p 2>/dev/null
for b in rb_w2 rb_w2p; do
echo "== $b"
for T in 2 16 28 36; do for bpc in 1 2; do timeout 60 /tmp/$b 1 $T 1048576 $bpc | head -1; done; done
done
python scripts/predict_overhead_probe.py 28
