import sys, os, time, threading
sys.path.insert(0, "image-to-video-i2v-attack_amd"); sys.path.insert(0, ".")
import torch
from i2v_amd import attacks, sign_attacks, video
import bench
K = int(sys.argv[1]); mt = sys.argv[2] if len(sys.argv) > 2 else "slowfast_resnet50"
dev = "cuda:0"
eng = attacks.get_engine(dev)
std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1)
jobs = []
for k in range(K):
    ori = bench.synthetic_clips(1, seed0=1000 + k).to(dev)
    gen = torch.Generator().manual_seed(77 + k)
    adv = (ori + (torch.randint(-10, 11, ori.shape, generator=gen).float() / 255 / std).to(dev)).contiguous()
    atk = sign_attacks.ILAF(video.VideoModel(mt, (32, 224, 224)), mt, engine=eng)
    jobs.append((atk, adv, ori, torch.cuda.Stream(device=dev)))
def run(j, reps):
    atk, adv, ori, st = j
    with torch.cuda.stream(st):
        for _ in range(reps):
            atk(adv, ori, torch.zeros(1, dtype=torch.long), ["v"])
for j in jobs: run(j, 1)
torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(j, reps)) for j in jobs]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(mt, "streams", K, "frames/s", round(K * reps * 32 / el, 1), "costs", [round(float(j[0].last_costs[-1]), 4) for j in jobs][:2])
