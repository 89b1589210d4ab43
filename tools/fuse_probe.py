import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/image-to-video-i2v-attack_amd")
import torch
from i2v_amd import attacks, graphs, weights
eng = attacks.get_engine("cuda:0")
g = graphs.build("resnet50", (224, 224))
sd = weights.synthetic_state_dict(g, 0)
for frames in (128, 32):
    net = eng.build_net(g, sd, [g.hooks[3]], frames)
    print(frames, "frames: fusion info (eligible fwd, bwd, fused fwd, bwd) =", net.fusion_info(), flush=True)
    net.close()
