"""Developer probe: the headline I2V attack with the batch split over K concurrent streams (threads, own attack
object and HIP stream each) vs one stream.  python tools/i2v_streams_probe.py <streams> <clips_total>"""
import sys, time, threading
sys.path.insert(0, "image-to-video-i2v-attack_amd"); sys.path.insert(0, ".")
import torch
from i2v_amd import attacks
import bench
K, total = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
eng = attacks.get_engine(dev)
per = total // K
jobs = []
for k in range(K):
    vid = bench.synthetic_clips(per, seed0=1000 + k * per).to(dev)
    atk = attacks.ImageGuidedFMDirection_Adam(["resnet50"], depth=3, step_size=0.005, steps=10, engine=eng)
    jobs.append((atk, vid, torch.cuda.Stream(device=dev)))
def run(j, reps):
    atk, vid, st = j
    with torch.cuda.stream(st):
        for _ in range(reps):
            atk(vid, torch.zeros(per, dtype=torch.long), [f"c{i}" for i in range(per)])
        st.synchronize()
for j in jobs: run(j, 1)
torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(j, reps)) for j in jobs]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("streams", K, "clips/stream", per, "frames/s", round(total * reps * 32 / el, 1))
