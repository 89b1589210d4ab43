"""Developer soak test (GPU box): random 3-D / 2-D max-pooling windows between two convolutions, forward + input gradient
through the C ABI against torch max_pool3d / autograd in float64.  python tools/soak_pool.py <seconds> [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-to-video-i2v-attack_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from i2v_amd import attacks, graphs, weights
from oracle import video_models as vm
from tests.test_gpu_video import write_hook_grads

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
eng = attacks.get_engine("cuda:0")
t_end, n, worst, ties = time.time() + budget, 0, 0.0, 0
while time.time() < t_end:
    kt, k = rnd.choice([1, 1, 2, 3]), rnd.choice([1, 2, 3, 3, 4])
    st, s = rnd.choice([1, 2, 2]), rnd.choice([1, 2, 2, 3])
    pt, p = rnd.choice([0, kt // 2]), rnd.choice([0, k // 2])
    T, H, W = rnd.choice([4, 6, 8, 9]), rnd.choice([9, 12, 16, 21]), rnd.choice([8, 12, 15, 28])
    if T + 2 * pt < kt or H + 2 * p < k or W + 2 * p < k or (kt == 1 and k == 1):
        continue
    C, b = rnd.choice([4, 8, 13, 32]), rnd.choice([1, 2])
    relu_a = rnd.random() < 0.7
    g = graphs.Graph("soakpool", (H, W), video=True)
    x = g.new_tensor(3, H, W, False, "input", T=T)
    g.input = x
    a = g.conv3d(x, C, (1, 3), (1, 1), (0, 1), "a.weight", bn="a_bn", relu=relu_a)
    pl = g.maxpool3d(a, (kt, k), (st, s), (pt, p))
    y = g.conv3d(pl, 8, (1, 1), (1, 1), (0, 0), "c.weight", bn="c_bn", relu=True)
    g.hooks[1] = y
    sd = weights.synthetic_state_dict(g, n)
    net = eng.build_net(g, sd, [y], b * T)
    xv = torch.randn(b, 3, T, H, W, dtype=torch.float64, generator=torch.Generator().manual_seed(n), requires_grad=True)

    def bn(t, pre):
        return F.batch_norm(t, sd[pre + ".running_mean"].double(), sd[pre + ".running_var"].double(),
                            sd[pre + ".weight"].double(), sd[pre + ".bias"].double(), False, 0.0, 1e-5)
    av = bn(F.conv3d(xv, sd["a.weight"].double(), None, 1, (0, 1, 1)), "a_bn")
    av = F.relu(av) if relu_a else av
    pv = F.max_pool3d(av, (kt, k, k), (st, s, s), (pt, p, p))
    yv = F.relu(bn(F.conv3d(pv, sd["c.weight"].double()), "c_bn"))
    net.forward(vm.to_frames(xv.detach()).float().to("cuda:0").contiguous())
    fy = vm.to_frames(yv.detach())
    got = net.save_hook(0, fy.shape[0]).cpu().double()
    e1 = float((got - fy).abs().max() / (fy.abs().max() + 1e-9))
    hg = torch.randn_like(yv)
    ref = vm.to_frames(torch.autograd.grad((yv * hg).sum(), xv)[0])
    write_hook_grads(net, [fy], [vm.to_frames(hg)])
    gx = torch.empty(b * T, 3, H, W, device="cuda:0")
    net.backward(gx)
    e2 = float((gx.cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-12))
    if e1 > 1e-4 or e2 > 1e-4:
        # ReLU ties (many equal zeros in a window: torch and the engine both take the first maximum in scan order, but
        # fp32 vs f64 can disagree on WHICH elements are zero) and last-bit gate flips are not errors
        fa = vm.to_frames(av.detach())
        ga = net.read_tensor(net.graph.nodes[0].dst, fa.shape[0]).cpu().double()
        flips = int(((fa > 0) != (ga > 0)).sum()) + int(((fy > 0) != (got > 0)).sum())
        # ... and so is a window whose two largest values swap order between fp32 and f64 (arg-max flip)
        ga5 = ga.reshape(b, T, C, H, W).permute(0, 2, 1, 3, 4)
        io = F.max_pool3d(av.detach(), (kt, k, k), (st, s, s), (pt, p, p), return_indices=True)[1]
        ie = F.max_pool3d(ga5, (kt, k, k), (st, s, s), (pt, p, p), return_indices=True)[1]
        flips += int((io != ie).sum())
        if flips and e1 <= 1e-4:
            ties += 1
        else:
            print("FAIL", dict(n=n, kt=kt, k=k, st=st, s=s, pt=pt, p=p, T=T, H=H, W=W, C=C, b=b, relu_a=relu_a), e1, e2)
            sys.exit(1)
    else:
        worst = max(worst, e1, e2)
    net.close()
    n += 1
print("pool soak ok:", n, "cases, worst relative error", worst, "; gate-flip cases:", ties)
