for spec in "nw 15" "nw 31" "nw 48" "sg 15" "sg 31" "sg 48" "sw 15" "sw 31" "sw 48"; do python profiles/bench_band_one.py $spec; done
echo "--- 4x8 for band 15"
for m in nw sg sw; do PMX_BSTRIP_SHAPE=4x8 python profiles/bench_band_one.py $m 15; done
echo "--- 8x8 for band 31"
for m in nw sg sw; do PMX_BSTRIP_SHAPE=8x8 python profiles/bench_band_one.py $m 31; done
