"""The compiled gfx950 code object of the stream kernels, checked without a GPU.

Why: the persistent kernels keep their whole pipeline state in registers on purpose.  Twice the compiler quietly moved a piece of
it to memory (a stack slot reloaded behind `s_waitcnt vmcnt(0)` every step; then an LDS slot behind `lgkmcnt(0)`): results stay
right, the 16-bit kernels ran 25-30 % slower.  The kernel descriptors say so in two numbers, so they are pinned
here: no private (scratch) segment, no VGPR spills, and exactly the LDS the `__shared__` arrays of the source ask for."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_metadata(tmp_path):
    lib = os.path.join(ROOT, "sparta_amd", "libsparta_amd.so")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(lib) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM binutils not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    # one offload bundle per kernel translation unit (k_*.hip), back to back in the section
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundle in libsparta_amd.so"
    kernels = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, co = str(tmp_path / ("bundle%d.bin" % n)), str(tmp_path / ("dev%d.co" % n))
        open(piece, "wb").write(blob[a:b])
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + piece, "--output=" + co], check=True)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
        for block in re.split(r"\n  - \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            kernels[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                             for k in ("private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "vgpr_count")}
    return kernels


def test_stream_kernels_keep_their_state_in_registers(tmp_path):
    kernels = _kernel_metadata(tmp_path)
    stream = {n: m for n, m in kernels.items() if "stream_kernel" in n or "direct_kernel" in n}
    assert len(stream) >= 6 + 2 + 32 + 16 + 12, sorted(stream)                 # fp32: 6 LDS-staged instantiations + the no-barrier kernel, 16-bit LDS-staged: 32, direct: 16
    for name, m in stream.items():
        assert m["private_segment_fixed_size"] == 0, (name, m)
        assert m["vgpr_spill_count"] == 0, (name, m)
        quad = ("h16_direct_kernelILi64ELb1E" in name or "h16_direct_kernelILi32ELb1E" in name) and "ELi64ELb1EEEvN10sparta_dev" in name      # <KP, two tiles, ..., WC = 64, SLAB>: four accumulators per wave
        if quad:
            assert m["vgpr_count"] <= 512, (name, m)                      # ONE workgroup per CU (the 16-bit plans' own choice): accumulators in the upper half of the file
        else:
            assert m["vgpr_count"] <= 256, (name, m)                      # two 256-thread workgroups per CU
        if "f32_direct_kernel" in name:                                   # 4 waves x 2 stages x 32 columns x (32 + 4) floats, private to each wave
            want = {4 * 2 * 32 * 36 * 4, 4 * 2 * 32 * 36 * 4 + 4 * 32 * 65 * 4}     # (+ the C ring of the CSTAGE instantiation: 4 waves x 32 columns x 65 floats)
        elif "h16_direct_kernel" in name:                                 # 4 waves x 2 stages x 32 columns x (KP + 8) 16-bit elements, private to each wave
            ring = 4 * 32 * 65 * 4                                        # the C ring of the CSTAGE instantiations (tiles of arbitrary height)
            want = {4 * 2 * 32 * (32 + 8) * 2, 4 * 2 * 32 * (32 + 8) * 2 + ring} if "ILi32E" in name else {4 * 2 * 32 * (64 + 8) * 2, 4 * 2 * 32 * (64 + 8) * 2 + ring}
            if "ELi64ELb0EEEvN10sparta_dev" in name or "ELi64ELb1EEEvN10sparta_dev" in name:                          # WC = 64: every wave holds the images of its two groups of 32 columns
                want = {4 * 2 * 64 * (32 + 8) * 2} if "ILi32E" in name else {4 * 2 * 64 * (64 + 8) * 2}
        elif "h16_stream_kernel" in name:                                 # 2 stages x (128 + 64) rows x (KP + 8) 16-bit elements
            want = {2 * (128 + 64) * (32 + 8) * 2} if "ILi32E" in name else {2 * (128 + 64) * (64 + 8) * 2}
        else:                                                             # fp32: 2 stages x (B panel [+4 pad when column-major] + 32 x 64 A slice) floats
            want = {2 * (128 * 36 + 32 * 64) * 4, 2 * (32 * 128 + 32 * 64) * 4}
        assert m["group_segment_fixed_size"] in want, (name, m, want)


def _disassemble(tmp_path):
    """text of llvm-objdump -d for every gfx950 code object in the library, split per kernel symbol"""
    lib = os.path.join(ROOT, "sparta_amd", "libsparta_amd.so")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(lib) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM binutils not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    kernels = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, co = str(tmp_path / ("dbundle%d.bin" % n)), str(tmp_path / ("ddev%d.co" % n))
        open(piece, "wb").write(blob[a:b])
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + piece, "--output=" + co], check=True)
        txt = subprocess.run([tools[2], "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        for block in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", txt):
            m = re.match(r"[0-9a-f]+ <([^>]+)>:", block)
            if m and not m.group(1).startswith((".", "$")) and "LBB" not in m.group(1):
                kernels[m.group(1)] = kernels.get(m.group(1), "") + block
    return kernels


def test_steady_state_of_the_one_tile_kernels_keeps_its_prefetch_and_its_valu_budget(tmp_path):
    """what cost these kernels their speed in the past shows in the ISA: a full `s_waitcnt vmcnt(0)` in the step (the
    prefetch distance is gone), accumulators or in-flight registers copied with `v_mov` at a join, pipeline state in scratch.  Pinned for the
    fp32 and 16-bit no-barrier kernels of the flagship shape: full waits only where the source has them (the C += path of the epilogue and the
    window swap), register moves within the epilogue's budget, step waits that leave the loads of the following steps in flight."""
    ks = _disassemble(tmp_path)
    f32 = [t for n, t in ks.items() if "vbs_spmm_f32_direct_kernelILb0E" in n]      # (the instantiation without the C ring: the flagship's)
    h16 = [t for n, t in ks.items() if "vbs_spmm_h16_direct_kernel" in n and "ILi32ELb0E" in n and "Lb0ELb0ELb1ELi32ELb0EEEvN10sparta_dev" in n]      # (<32, one tile, *, *, CSTAGE = false, DEEP = false, TAIL = true>)
    h16_deep = [t for n, t in ks.items() if "vbs_spmm_h16_direct_kernel" in n and "ILi32ELb0E" in n and "Lb0ELb1ELb0ELi32ELb0EEEvN10sparta_dev" in n]  # (the same seven steps ahead, TAIL = false: the flagship's)
    h16_wide = [t for n, t in ks.items() if "vbs_spmm_h16_direct_kernel" in n and "ILi32ELb0E" in n and "Lb0ELb0ELb0ELi64ELb0EEEvN10sparta_dev" in n]  # (64-column waves, TAIL = false: the flagship's)
    assert len(f32) == 2 and len(h16) >= 4 and len(h16_deep) == 2 and len(h16_wide) == 4, sorted(ks)[:8]      # (f32: TAIL = true / false)
    for txt in h16_deep:
        ins = [l.split("//")[0].strip() for l in txt.splitlines() if l.startswith(("\t", " "))]
        ins = [i for i in ins if i]
        assert not any(i.startswith("scratch_") for i in ins)
        steps = sum(i.startswith("v_mfma") for i in ins) / 2          # a round of 8 + up to 7 peeled
        assert steps == 15, steps
        n_mov = sum(2 if i.startswith("v_mov_b64") else 1 for i in ins if i.startswith("v_mov_b"))
        assert n_mov <= 32 * steps, (n_mov, steps)
        waits = [int(m.group(1)) for i in ins for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", i)] if m]
        # the step waits of the round leave the loads of five steps and more in flight (two waits per step: one per LDS write)
        assert sum(w >= 20 for w in waits) >= 2 * 8, waits
        assert sum(w == 0 for w in waits) <= 1.5 * steps, waits
    for txt, n_mfma_step, loads_per_step, mov_budget in [(t, 16, 9, 44) for t in f32] + [(t, 2, 4, 34) for t in h16] + [(t, 4, 6, 68) for t in h16_wide]:
        ins = [l.split("//")[0].strip() for l in txt.splitlines() if l.startswith(("\t", " "))]
        ins = [i for i in ins if i]
        assert not any(i.startswith("scratch_") for i in ins)
        n_mfma = sum(i.startswith("v_mfma") for i in ins)
        steps = n_mfma / n_mfma_step                                   # unrolled step bodies in the kernel (loop of 4 + up to 3 peeled + prologue pieces)
        assert 7 <= steps <= 12, steps
        # register moves: 16 copies + 16 clears per epilogue body (v_mov_b64 counts two), nothing per step
        n_mov = sum(2 if i.startswith("v_mov_b64") else 1 for i in ins if i.startswith("v_mov_b"))
        assert n_mov <= mov_budget * steps, (n_mov, steps)            # (28 per body with TAIL, 32 without: measured 1 % faster all the same; two accumulators: twice)
        # full waits: only in the epilogue bodies' C += path (one per body)
        n_full = sum(bool(re.match(r"s_waitcnt vmcnt\(0\)", i)) for i in ins)
        assert n_full <= (1.5 if mov_budget < 40 else 2.5) * steps, (n_full, steps)     # (13 in 11 bodies today; two column groups: two C += paths per body)
        # the waits in front of the LDS writes of a step leave at least one step's loads in flight
        waits = [int(m.group(1)) for i in ins for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)", i)] if m]
        assert sum(w >= loads_per_step for w in waits) >= 4 * steps, (waits, steps)


def test_resident_column_kernels_keep_their_sums_in_registers(tmp_path):
    """k_colres.hip: the sums of a wave's slices live in a statically indexed register array (`acc[SL][NC]`) through the stream of A.  Written the obvious way -- conditional stores
    at a slice's end, or the next set's columns of B held next to them -- the compiler moves the array to scratch: eight instantiations (1..4 columns x
    values / unit image), no private segment, no spills, at most 128 registers (four waves per SIMD: one 1024-thread workgroup per CU), no static LDS (the launch sizes it)."""
    kernels = _kernel_metadata(tmp_path)
    cr = {n: m for n, m in kernels.items() if "colres_kernel" in n}
    assert len(cr) == 8, sorted(cr)
    for name, m in cr.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (name, m)
        assert m["vgpr_count"] <= 128, (name, m)
        assert m["group_segment_fixed_size"] == 0, (name, m)


def test_column_compacted_tile_kernel_fits_three_workgroups_per_cu(tmp_path):
    """k_union.hip: three workgroups per CU is the point of its two-stage pipeline (profiles/r5/lab_union_stages.txt): 50 KB of LDS (two stages of the tallest tile type: 8 KB of A + 1 KB of tail pairs +
    16 KB of B), at most 168 registers (three waves per SIMD), nothing in scratch; its panel loads are one LDS-direct load per 1 KB piece; and the matrix instruction is the 16-row one (16 RT per step of the body of
    RT row tiles, RT = 1..4: a 48-row cluster pays for three row tiles, not four)."""
    kernels = _kernel_metadata(tmp_path)
    un = {n: m for n, m in kernels.items() if "vbs_union_f32_kernel" in n}
    assert len(un) == 1, sorted(un)
    for name, m in un.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (name, m)
        assert m["vgpr_count"] <= 168, (name, m)
        assert m["group_segment_fixed_size"] == 2 * (2 * 4096 + 1024 + 32 * 512) and 3 * m["group_segment_fixed_size"] <= 160 * 1024, (name, m)
    txt = [t for n, t in _disassemble(tmp_path).items() if "vbs_union_f32_kernel" in n]
    assert len(txt) == 1
    ins = [l.split("//")[0].strip() for l in txt[0].splitlines() if l.startswith(("\t", " "))]
    # per step and body: the wave's pieces of the slice of A (buffer loads) and its four 1 KB pieces of the panel of B -- ONE LDS-direct load each, per-lane source addresses
    # (global_load_lds_dwordx4), not one per row under half an exec mask: an LDS-direct load costs its wave 100-185 cycles of issue
    a_loads = [i for i in ins if i.startswith("buffer_load_dwordx4") and i.endswith("lds")]
    b_loads = [i for i in ins if i.startswith("global_load_lds_dwordx4")]
    assert len(a_loads) >= 4 and len(b_loads) == 4 * 4 * 2 and not any(i.startswith("scratch_") for i in ins), (len(a_loads), len(b_loads))   # (four bodies x four pieces, prologue + loop)
    assert sum(i.startswith("v_mfma_f32_16x16x4") for i in ins) == 16 * (1 + 2 + 3 + 4)          # the four tile types' step bodies
    assert not any(i.startswith("v_mfma_f32_32x32x2") for i in ins)
