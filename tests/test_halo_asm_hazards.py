"""Static check of the hand-scheduled halo-tile kernels (conv_imggrad_halo, conv_stem_halo, conv_stem64_halo).

Their weight fragments arrive by inline-asm `buffer_load_dword` into a register ring and every `s_waitcnt vmcnt(N)` is written
by hand: the compiler does not know that a load's destination register is still in flight.  Round 5 lost an evening to exactly
that -- in the fully unrolled wide-stem kernel the ring's run-out loads were dead values, the register allocator pointed all of
them at one scratch register and handed it to an accumulator, and a late load landed in a live accumulator.  This test compiles
the kernels for gfx950 (device code only, no GPU needed), walks each kernel's instruction stream with the wave's VMEM queue
modelled as the hardware keeps it (in order; `s_waitcnt vmcnt(N)` retires all but the youngest N), and fails if ANY instruction
reads or writes a VGPR that an in-flight asm load still targets.  It is a linear walk (loop back-edges are not followed): the
kernels are written so that every loop body begins and ends with the queue in the state the walk sees."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def scan(lines):
    """-> list of (line number, instruction, pending (register, issued at line)) hazards"""
    pending, hits = [], []                       # VMEM queue, oldest first: (destination VGPR or -1 for LDS-DMA, line)
    for ln, raw in enumerate(lines, 1):
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        m = re.match(r"s_waitcnt\b.*\bvmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n < len(pending) else pending
            if n == 0:
                pending = []
            continue
        if t.startswith(("global_store", "buffer_store", "flat_store")):
            break                                 # the epilogue: compiler-managed memory traffic from here on
        live = {r for r, _ in pending if r >= 0}
        if t.startswith("buffer_load_dword") and " lds" in t:
            if vregs(t) & live:
                hits.append((ln, t, sorted(vregs(t) & live)))
            pending.append((-1, ln))
            continue
        m = re.match(r"buffer_load_dword v(\d+), (.*)", t)
        if m:
            dst = int(m.group(1))
            if (vregs(m.group(2)) | {dst}) & live:
                hits.append((ln, t, sorted((vregs(m.group(2)) | {dst}) & live)))
            pending.append((dst, ln))
            continue
        if vregs(t) & live:
            hits.append((ln, t, sorted(vregs(t) & live)))
    return hits


def test_scanner_sees_a_load_landing_in_a_live_register():
    bad = """
        buffer_load_dword v68, v79, s[16:19], s1 offen
        v_mfma_f32_16x16x4_f32 v[68:71], v75, v101, v[30:33]
        s_waitcnt vmcnt(0)
    """.splitlines()
    good = """
        buffer_load_dword v68, v79, s[16:19], s1 offen
        buffer_load_dword v69, v79, s[16:19], s2 offen
        v_mfma_f32_16x16x4_f32 v[30:33], v75, v101, v[30:33]
        s_waitcnt vmcnt(1)
        v_mfma_f32_16x16x4_f32 v[30:33], v68, v101, v[30:33]
        s_waitcnt vmcnt(0)
        v_mfma_f32_16x16x4_f32 v[30:33], v69, v101, v[30:33]
    """.splitlines()
    assert scan(bad) and not scan(good)


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc: the kernels cannot be compiled here")
def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path):
    asm = tmp_path / "halo.s"
    # the product's own translation unit of these kernels, with the experimental wide-stem kernel compiled in as well (device pass only)
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++20", "-Wno-unused-value",
           "-Wno-unused-result", "-x", "hip", "-DI2V_EXPERIMENTAL", "--cuda-device-only", "-S",
           os.path.join(ROOT, "image-to-video-i2v-attack_amd", "csrc", "i2v_conv_stems.hip"), "-o", str(asm)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = asm.read_text().splitlines()
    kernels = {}
    name = None
    for line in text:
        m = re.match(r"^(_Z\d+conv_(?:imggrad_halo|stem_halo|stem64_halo)\w*):", line)
        if m:
            name = m.group(1); kernels[name] = []
            continue
        if name is not None:
            kernels[name].append(line)
            if "s_endpgm" in line:
                name = None
    assert len(kernels) >= 9, sorted(kernels)            # six image-gradient instantiations, the narrow stem, two wide-stem ones
    for k, body in kernels.items():
        loads = sum(1 for l in body if re.match(r"\s*buffer_load_dword v\d+", l) and " lds" not in l)
        assert loads >= 8, (k, loads)                    # the ring is there (the walk would pass vacuously on an empty stream)
        hits = scan(body)
        assert not hits, (k, hits[:5])
        # M0 (the LDS-DMA base) is written by the kernels' own asm and is not on its clobber list (the compiler rejects reserved registers
        # there): that is sound only while nothing else in the kernel touches M0 -- every mention must be one of those writes, each
        # followed by its `s_nop` and its `buffer_load ... lds`
        ins = [l.split(";")[0].strip() for l in body]
        ins = [t for t in ins if t and not t.startswith(".") and not t.endswith(":")]
        dma = 0
        for i, t in enumerate(ins):
            if re.search(r"\bm0\b", t):
                assert re.match(r"s_mov_b32 m0, s\d+$", t), (k, t)
                assert ins[i + 1].startswith("s_nop") and ins[i + 2].startswith("buffer_load_dword") and ins[i + 2].endswith(" lds"), (k, ins[i:i + 3])
                dma += 1
        assert dma == sum(1 for t in ins if t.startswith("buffer_load_dword") and t.endswith(" lds")) and dma >= 5, (k, dma)
