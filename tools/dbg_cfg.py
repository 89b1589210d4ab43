import os, sys
ROOT='/root/repo'
for p in (os.path.join(ROOT,'image-to-video-i2v-attack_amd'), ROOT): sys.path.insert(0,p)
os.environ["I2V_AUTOTUNE"]="0"; os.environ["I2V_FORCE_CFG"]=sys.argv[1] if len(sys.argv)>1 else "4"
import torch
from i2v_amd import attacks, graphs, weights
from oracle import restate
eng=attacks.get_engine("cuda:0")
for model, depth in (("resnet", 3), ("squeezenet", 2), ("densenet121", 2)):
    g = graphs.build_tiny(model, (64, 64)); sd = weights.synthetic_state_dict(g, 3); hooks=[g.hooks[depth]]
    net = eng.build_net(g, sd, hooks, 3)
    onet = restate.OracleNet(g, sd, hooks, dtype=torch.float64)
    torch.manual_seed(4); x = torch.randn(3,3,64,64)
    onet.forward(x.double()); net.forward(x.to("cuda:0"))
    tg = net.graph
    for nd in tg.nodes:
        got = net.read_tensor(nd.dst, 3).cpu().double(); ref = onet.tensor(nd.dst)
        err = (got-ref).abs()
        if err.max() > 1e-4*ref.abs().max():
            bad = (err > 1e-4*ref.abs().max()).nonzero()
            print(model, tg.tensors[nd.dst].name, tuple(ref.shape), "max err", float(err.max()), "ref max", float(ref.abs().max()), "nbad", len(bad), "first", bad[:6].tolist())
            break
    else:
        print(model, "all layers ok")
    net.close()
