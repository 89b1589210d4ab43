import sys, time, os
sys.path.insert(0,'image-to-video-i2v-attack_amd'); sys.path.insert(0,'.')
import torch
from i2v_amd import graphs, weights
from oracle import restate
g = graphs.build("resnet50",(224,224))
net = restate.OracleNet(g, weights.synthetic_state_dict(g,0), [g.hooks[3]])
for thr in [16, 32, 64, 128]:
    torch.set_num_threads(thr)
    for nf in [8, 32]:
        x = torch.randn(nf,3,224,224)
        t0=time.time(); f = net.forward(x); t1=time.time()
        gx = net.backward([torch.randn_like(f[0])]); t2=time.time()
        print(f"threads {thr} frames {nf}: fwd {t1-t0:.2f}s bwd {t2-t1:.2f}s", flush=True)
