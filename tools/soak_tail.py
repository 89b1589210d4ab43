"""Developer soak test (GPU box) for the 64x64 tile's variants: random three-layer image nets run with the plain 64x64 configuration and
with a variant forced onto every eligible launch (no autotuning): features and input gradient must agree BIT FOR BIT.
    python tools/soak_tail.py <seconds> [seed] [tail|halo|dc|nt|fuse|fusehalo]
  tail: the tail split (conv_igemm_tail, I2V_FORCE_CFG = 3 | 32) on nets large enough to leave a remainder over the 256 CUs;
  halo: halo staging (conv_igemm_halo, 3 | 16) on planes 14 / 28 / 56 wide, any height, few or many frames;
  dc:   two K chunks per barrier (conv_igemm_dc, 3 | 64; round 4), channel counts that are multiples of 32;
  fuse / fusehalo: the fused 3x3 -> pointwise pair (conv_fused_kernel, I2V_FORCE_FUSE = 1 / 2; round 4): a 64- or 128-channel 3x3
        layer followed by a pointwise layer of >= 64 channels, against the two-launch plan."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-to-video-i2v-attack_amd")); sys.path.insert(0, ROOT)
os.environ["I2V_AUTOTUNE"] = "0"
import torch
from i2v_amd import attacks, graphs, weights
from tests.test_gpu_video import write_hook_grads

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "tail"
variant = 3 | {"halo": 16, "tail": 32, "dc": 64, "nt": 128}.get(mode, 0)           # nt: streaming epilogue stores (round 4)
fuse = {"fuse": "1", "fusehalo": "2"}.get(mode)
rnd = random.Random(seed)
eng = attacks.get_engine("cuda:0")
t_end, n, split_cases, fused_pairs = time.time() + budget, 0, 0, 0
while time.time() < t_end:
    H = rnd.choice([14, 20, 28, 33, 56]); W = rnd.choice([14, 28, 56] if mode in ("halo", "fusehalo") else [14, 24, 28, 40, 56])
    c1 = rnd.choice([16, 32, 48, 64]); c2 = rnd.choice([16, 32, 64, 80, 128, 256])
    k = rnd.choice([1, 3])
    frames = rnd.choice([1, 2, 3, 5, 8, 24] if mode in ("halo", "dc", "nt", "fuse", "fusehalo") else [24, 40, 64, 96, 128])
    if mode == "dc":
        c1 = rnd.choice([32, 64, 96]); c2 = rnd.choice([64, 96, 128, 256])
    if fuse:            # a (3 -> c1) -> b (3x3, c1 -> 64 | 128: the fused pair's first half) -> c (1x1 -> c3 >= 64, optional residual)
        c1 = rnd.choice([32, 64, 128]); c2 = rnd.choice([64, 128]); k = 3
    # how many launches would be split?  (64-pixel tiles x ceil(C / 64) channel tiles, remainder over 256 in (0, 104])
    def eligible(C):
        t = ((frames * H * W + 63) // 64) * ((C + 63) // 64)
        return t >= 512 and 0 < t % 256 <= 104 and (t % 256) // ((C + 63) // 64) > 0
    split_cases += int(eligible(c1) or eligible(c2))
    g = graphs.Graph("soak_tail", (H, W))
    x = g.new_tensor(3, H, W, False, "input")
    g.input = x
    a = g.conv(x, c1, 3, 1, 1, "a.weight", bn="a_bn", relu=True)
    b = g.conv(a, c2, k, 1, k // 2, "b.weight", bn="b_bn", relu=rnd.random() < 0.7)
    c3 = rnd.choice([64, 96, 128, 256]) if fuse else c1
    c = g.conv(b, c3, 1, 1, 0, "c.weight", bn="c_bn", relu=True, residual=a if (c3 == c1 and rnd.random() < 0.5) else None)
    g.hooks[1] = c
    sd = weights.synthetic_state_dict(g, n)
    xin = torch.randn(frames, 3, H, W, generator=torch.Generator().manual_seed(n)).to("cuda:0")
    outs = []
    for cfg in (3, variant):
        os.environ["I2V_FORCE_CFG"] = str(cfg)
        if fuse:
            os.environ["I2V_FORCE_FUSE"] = fuse if len(outs) else "0"
        net = eng.build_net(g, sd, [c], frames)
        if fuse and len(outs):
            fused_pairs += sum(net.fusion_info()[2:])
        net.forward(xin)
        f = net.save_hook(0, frames).cpu()
        write_hook_grads(net, [f], [torch.randn(f.shape, generator=torch.Generator().manual_seed(n + 1))])
        gx = torch.empty(frames, 3, H, W, device="cuda:0")
        net.backward(gx)
        outs.append((f, gx.cpu()))
        net.close()
    if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
        print("FAIL", dict(n=n, H=H, W=W, c1=c1, c2=c2, k=k, frames=frames))
        sys.exit(1)
    n += 1
print(mode, "soak ok:", n, "nets" + (", %d with at least one split launch" % split_cases if mode == "tail" else "") +
      (", %d fused pairs executed" % fused_pairs if fuse else ""), "-- all bit-identical")
