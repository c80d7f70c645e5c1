set -e
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWALK2 scripts/rowreg_bench.hip"
$B -o /tmp/rb2 2>/dev/null
for f in NOIDX NOCMP CMPONLY IDXONLY NOADDC NOSMEM NOFMA NODS; do $B -DEXP_$f -o /tmp/rb2_$f 2>/dev/null; done
echo "== pure walk: 24 resident trees walked 40 times per tile (960 trees), 2^18 rows"
for bpc in 1 2 3; do timeout 60 /tmp/rb2 1 24 262144 $bpc 16 40; done
for f in NOIDX NOCMP CMPONLY IDXONLY NOADDC NOSMEM NOFMA NODS; do echo $f; timeout 60 /tmp/rb2_$f 1 24 262144 2 16 40; done
echo "== grouped"
for TT in 8 16; do timeout 120 /tmp/rb2 2 1000 1048576 2 $TT; done
for T in 16 28 36; do timeout 60 /tmp/rb2 1 $T 1048576 2; done
