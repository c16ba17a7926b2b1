for sh in 0,0,1 8,0,2 8,0,4 8,0,8; do for ov in 0 1 2; do echo "== shard=$sh overlap=$ov"; timeout -k 10 120 python tools/gpu_frames.py scene=obj frames=5 shard=$sh overlap=$ov | tail -2; done; done
for sc in blob6 hf708; do for ov in 0 1 2; do echo "== $sc overlap=$ov"; timeout -k 10 200 python tools/gpu_frames.py scene=$sc frames=4 overlap=$ov | tail -2; done; done
