# (round 5: variants 31 / 36 / 37 were ablation branches inside the production kernels' source; they exist up to commit 7317048)
for v in 25 36 37 31; do for shp in "42000 4096 1024" "42000 1024 4096"; do timeout 100 python tools/gemm_one.py $v $shp 2>&1 | grep -E "variant|rror" | tail -2; done; done; true
