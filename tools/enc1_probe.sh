D=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT)")
F=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT & ~b.TUNE_ATTN_ENC_F)")
for t in $D $F; do for b in 1 2 4; do echo "tuning $t batch $b: $(WH_TUNING=$t D1_BATCH=$b D1_STEPS=20 timeout 200 python tools/decode1_prof.py 2>&1 | grep 'encode batch')"; done; done
