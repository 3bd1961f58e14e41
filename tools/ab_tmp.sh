export SR3_LIB=3d-super-resolution-face-reconstruction_amd/libsr3hip_exp.so
for d in 0 1 2 4 3 5 6 7; do
SR3_ATTN_DBG=$d python tools/step_profile.py --batch 64 --steps 10 --image-size 128 2>&1 | grep "event-timed" | sed "s/.*attention/[dbg $d] attention/" | cut -c1-60
done
