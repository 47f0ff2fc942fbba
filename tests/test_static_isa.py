"""Static checks on the SHIPPED gfx950 code object (CPU suite: llvm-objdump, no GPU).

wblock36_mfma.h hides 7 / 15 MFMAs of every block inside one `asm volatile`; the compiler's hazard recogniser sees only
the block's last MFMA (its own builtin) -- and since round 5 NONE of the 16 of the 128-channel instance's blocks
(LAST_IN_ASM: this test is the gate for that switch).  Whether an accumulator written by an MFMA *inside* the asm is old enough when
the first non-MFMA instruction touches it rests on a timing argument (wblock36_mfma.h, fpc_mfma_step): this test turns
that argument into a build-time fact.  It disassembles every `wblock36_kernel` instance of feature-point-cnn_amd/lib/
libfpc.so (and of `wblock36_dust_kernel`, the same body with the detector's 65th channel) and checks

  (i)  MFMA result hazards: `v_mfma_f32_16x16x4_f32` is an 8-pass XDL instruction; a VALU / LDS / VMEM / accvgpr
       instruction that reads or overwrites its destination needs >= 11 wait states after the MFMA's issue (CDNA3 ISA
       guide 4.5 "XDL write VGPR -> VALU read/write, VMEM/LDS/FLAT read": passes + 3; LLVM's GCNHazardRecognizer
       GFX940_XDL_N_PassWriteVgprVALU*WaitStates).  Wait states are counted the way the hardware spends them, as a LOWER
       bound on elapsed issue cycles: every instruction 1, `s_nop k` k + 1, and an MFMA cannot issue earlier than 8 wait
       states (its 8 passes) after the previous MFMA.  s_waitcnt stalls only add to that, so a pass here is safe.
       Straight-line order and every backward branch (loop back edges: tail of the body followed by its head) are scanned.
  (ii) the spill facts DESIGN.md section 3.1 states: per instance `.sgpr_spill_count`, `.vgpr_spill_count`,
       `.private_segment_fixed_size` and the number of scratch instructions -- and that NONE of the spill traffic
       (v_readlane / v_writelane / scratch_*) sits inside the chunk loop (the MFMA-dense loop every tile runs
       nchunk / 2 times).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "feature-point-cnn_amd", "lib", "libfpc.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# What DESIGN.md section 3.1 ("Spills, as the code object has them") states.  SGPR spills are v_writelane / v_readlane
# pairs of uniform values (tile coordinates, row offsets of the tail's loads and stores); the VGPR spills of the
# 128-channel instances are 4 registers around the tile loop's head.  Update BOTH places when the kernel changes.
SPILLS = {
    # instance (NB, TYT, TXT[, "dust"]): (sgpr_spill_count max, vgpr_spill_count max, private_segment_fixed_size max)
    (1, 4, 4): (80, 0, 0),
    (1, 2, 8): (80, 0, 0),
    (2, 4, 4): (96, 4, 16),
    (2, 2, 8): (96, 4, 16),
    # wblock36_dust_kernel<TYT, TXT> (the detector's 64 + 1 channels: NB = 1 and the dustbin channel on the VALU)
    (1, 4, 4, "dust"): (184, 0, 0),
    (1, 2, 8, "dust"): (184, 0, 0),
    # wblock36p_kernel<TYT, TXT, RING> (round 5: the 64-channel instance at two waves per SIMD; accumulators in VGPRs, no AGPR
    # named anywhere -- with one, the compiler splits the wave's 256 registers 128 / 128 and spills 137)
    ("p", 4, 4): (48, 0, 0),
    ("p", 2, 8): (48, 0, 0),
    # wblock36p_dust_kernel<TYT, TXT, RING> (round 5: the paired instance + the detector's 65th channel): the only instance
    # with anything in scratch -- 7 / 9 registers saved around a role's chunk loop, once per TILE (checked below: nothing of
    # it inside a chunk loop)
    ("p", 4, 4, "dust"): (144, 12, 48),
    ("p", 2, 8, "dust"): (144, 12, 48),
}


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.skip("libfpc.so not built")
    if not all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")):
        pytest.skip("no LLVM tools under %s (a box without ROCm): the code object cannot be disassembled" % LLVM)
    d = tmp_path_factory.mktemp("isa")
    lib = os.path.join(d, "libfpc.so")
    shutil.copy(LIB, lib)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], stdout=subprocess.DEVNULL, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    assert len(co) == 1, os.listdir(d)
    co = os.path.join(d, co[0])
    dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co]).decode()
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    funcs, cur = {}, None
    for ln in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur and ln.strip() and not ln.startswith("Disassembly"):
            ins = ln.split("//")[0].strip()
            if ins:
                addr = int(ln.split("//")[1].split(":")[0].strip(), 16) if "//" in ln else None
                funcs[cur].append((addr, ins))
    meta = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                      for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
        meta[name]["agpr_count"] = int(blk.split()[0])
    return funcs, meta


def _instances(funcs):
    out = {}
    for name in funcs:
        m = re.match(r"_ZN3fpc15wblock36_kernelILi(\d+)ELi(\d+)ELi(\d+)EEEvNS_10WBlockArgsE$", name)
        if m:
            out[tuple(int(v) for v in m.groups())] = name
        m = re.match(r"_ZN3fpc20wblock36_dust_kernelILi(\d+)ELi(\d+)EEEvNS_10WBlockArgsE$", name)
        if m:
            out[(1,) + tuple(int(v) for v in m.groups()) + ("dust",)] = name
        m = re.match(r"_ZN3fpc16wblock36p_kernelILi(\d+)ELi(\d+)ELi(\d+)EEEvNS_10WBlockArgsE$", name)
        if m:
            out[("p",) + tuple(int(v) for v in m.groups()[:2])] = name
        m = re.match(r"_ZN3fpc21wblock36p_dust_kernelILi(\d+)ELi(\d+)ELi(\d+)EEEvNS_10WBlockArgsE$", name)
        if m:
            out[("p",) + tuple(int(v) for v in m.groups()[:2]) + ("dust",)] = name
    return out


_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))")


def _regs(operand_text):
    """set of ('a'|'v', index) named in an operand string."""
    out = set()
    for m in _REG.finditer(operand_text):
        f = m.group(1)
        if m.group(4) is not None:
            out.add((f, int(m.group(4))))
        else:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _touches_vector_registers(op):
    return op.startswith(("v_", "ds_", "buffer_", "global_", "flat_", "scratch_"))


def _scan(seq, need=11, mfma_passes=8):
    """seq: list of instruction strings in issue order -> list of (index, instruction, writer index, wait states)."""
    bad = []
    t = 0
    last_mfma_t = -10 ** 9
    pending = {}          # register -> (issue time of the MFMA that writes it, index)
    for i, ins in enumerate(seq):
        op = ins.split()[0]
        rest = ins[len(op):]
        if op.startswith("v_mfma"):
            t = max(t + 1, last_mfma_t + mfma_passes)
            last_mfma_t = t
            dst = rest.split(",")[0]
            # (an MFMA reading another MFMA's result as SrcC / A / B is interlocked by the hardware: not checked here)
            for r in _regs(dst):
                pending[r] = (t, i)
            continue
        if op == "s_nop":
            t += int(rest.strip(), 0) + 1
            continue
        t += 1
        if _touches_vector_registers(op):
            for r in _regs(rest):
                if r in pending:
                    t0, j = pending[r]
                    if t - t0 < need:
                        bad.append((i, ins, j, t - t0))
                    else:
                        del pending[r]
    return bad


def test_wblock36_uses_only_the_8_pass_mfma(code_object):
    funcs, _ = code_object
    inst = _instances(funcs)
    assert set(inst) == set(SPILLS), sorted(inst)
    for key, name in inst.items():
        ops = {i.split()[0] for _, i in funcs[name] if i.startswith("v_mfma")}
        assert ops == {"v_mfma_f32_16x16x4_f32"}, (key, ops)      # the wait-state figure below is this instruction's


def test_no_instruction_touches_an_mfma_result_too_early(code_object):
    """(i) of the module docstring, for every wblock36_kernel instance."""
    funcs, _ = code_object
    for key, name in _instances(funcs).items():
        body = funcs[name]
        seq = [i for _, i in body]
        bad = _scan(seq)
        assert not bad, "%s: %s" % (key, [(seq[j], ins, ws) for _, ins, j, ws in bad[:5]])
        # loop back edges: the 40 instructions in front of a backward branch, then the 40 at its target
        addr_to_idx = {a: k for k, (a, _) in enumerate(body) if a is not None}
        first = body[0][0]
        n_back = 0
        for k, (a, ins) in enumerate(body):
            if not ins.startswith(("s_cbranch", "s_branch")):
                continue
            # llvm-objdump prints the target as `<symbol+0xOFFSET>` in the comment we stripped, or a simm16 in words
            simm = int(ins.split()[1], 0)
            if simm >= 0x8000:
                simm -= 0x10000
            tgt = (a + 4 + 4 * simm) if a is not None else None
            if tgt is None or tgt not in addr_to_idx or addr_to_idx[tgt] > k:
                continue
            n_back += 1
            head = addr_to_idx[tgt]
            bad = _scan(seq[max(0, k - 40):k] + seq[head:head + 40])
            assert not bad, "%s back edge at %d: %s" % (key, k, bad[:3])
        assert n_back >= 2, (key, n_back, hex(first or 0))      # the chunk loop and the tile loop at least


def _back_edges(body):
    addr_to_idx = {a: k for k, (a, _) in enumerate(body) if a is not None}
    out = []
    for k, (a, ins) in enumerate(body):
        if ins.startswith(("s_cbranch", "s_branch")) and a is not None:
            simm = int(ins.split()[1], 0)
            if simm >= 0x8000:
                simm -= 0x10000
            tgt = a + 4 + 4 * simm
            if tgt in addr_to_idx and addr_to_idx[tgt] < k:
                out.append((addr_to_idx[tgt], k))
    return out


def _chunk_loop(body):
    """(first, last, MFMAs) of the innermost loop (no other back edge inside) that holds the most MFMAs."""
    edges = _back_edges(body)
    best = None
    for head, k in edges:
        if any(h2 >= head and k2 <= k and (h2, k2) != (head, k) for h2, k2 in edges):
            continue
        n = sum(1 for _, i in body[head:k] if i.startswith("v_mfma"))
        if best is None or n > best[2]:
            best = (head, k, n)
    return best


def test_the_scanner_sees_a_planted_hazard():
    """The checker itself: a reader 3 wait states behind an asm-style MFMA is reported, the same reader behind a second
    MFMA (8 passes) + s_nop 1 is not; a store of the accumulator and an accvgpr read count as readers."""
    blk = ["v_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]"]
    assert _scan(blk + ["s_nop 1", "v_accvgpr_read_b32 v5, a2"])
    assert _scan(blk + ["s_nop 1", "ds_write_b32 v9, a1"])                     # LDS data straight from an AGPR: a reader too
    assert _scan(blk + ["s_nop 1", "v_add_f32_e32 v7, v7, v8"]) == []         # unrelated registers
    # a nop in front of the next MFMA is absorbed by the 8 passes the matrix pipe is busy anyway: 8 + 1 = 9 wait states
    assert _scan(blk + ["s_nop 1", "v_mfma_f32_16x16x4_f32 a[4:7], v1, v2, a[4:7]", "v_accvgpr_read_b32 v5, a2"])
    ok = blk + ["v_mfma_f32_16x16x4_f32 a[4:7], v1, v2, a[4:7]", "s_nop 1", "v_accvgpr_read_b32 v5, a2"]
    assert _scan(ok) == []                                                     # 8 + 2 + 1 = 11 wait states
    assert _scan(blk + ["v_mfma_f32_16x16x4_f32 a[4:7], v1, v2, a[4:7]", "v_accvgpr_read_b32 v5, a6"])   # the second one's result
    assert _scan(["v_mfma_f32_16x16x4_f32 v[10:13], v1, v2, v[10:13]", "s_nop 7", "v_max_f32_e32 v11, v11, v0"])
    assert _scan(["v_mfma_f32_16x16x4_f32 v[10:13], v1, v2, v[10:13]", "s_nop 7", "s_nop 1", "v_max_f32_e32 v11, v11, v0"]) == []


def test_spills_are_what_design_md_says_and_outside_the_chunk_loop(code_object):
    """(ii) of the module docstring."""
    funcs, meta = code_object
    for key, name in _instances(funcs).items():
        m = meta[name]
        s_max, v_max, p_max = SPILLS[key]
        assert m["sgpr_spill_count"] <= s_max, (key, m)
        assert m["vgpr_spill_count"] <= v_max, (key, m)
        assert m["private_segment_fixed_size"] <= p_max, (key, m)
        body = funcs[name]
        n_scratch = sum(1 for _, i in body if i.startswith("scratch_"))
        assert n_scratch <= (0 if v_max == 0 else 12 if key[0] != "p" else 40), (key, n_scratch)
        loop = _chunk_loop(body)
        assert loop is not None, key
        head, tail, n_mfma = loop
        # two chunks per trip: 2 x 36 positions x 4 k-steps x NB channel blocks (the paired kernel: a wave's 18 positions,
        # one such loop per role)
        assert n_mfma == (2 * 18 * 4 if key[0] == "p" else 2 * 36 * 4 * key[0]), (key, n_mfma)
        if key[0] == "p":
            assert m["agpr_count"] == 0 and m["vgpr_count"] <= 256, (key, m)
        inside = [i for _, i in body[head:tail] if i.startswith(("v_readlane", "v_writelane", "scratch_"))]
        if key[0] == "p":
            # BOTH role loops (the transforming waves' and the requesting waves'): every innermost loop of 144 MFMAs
            edges = _back_edges(body)
            loops = [(h, k) for h, k in edges
                     if sum(1 for _, i in body[h:k] if i.startswith("v_mfma")) == 144
                     and not any(h2 >= h and k2 <= k and (h2, k2) != (h, k) and
                                 sum(1 for _, i in body[h2:k2] if i.startswith("v_mfma")) for h2, k2 in edges)]
            assert len(loops) == 2, (key, loops)
            for h, k in loops:
                assert not [i for _, i in body[h:k] if i.startswith("scratch_")], (key, h, k)
        if "dust" in key and key[0] == "p":
            assert not [i for i in inside if i.startswith("scratch_")], (key, inside[:5])
        elif "dust" in key:
            # the dust instance reloads the LDS-DMA's M0 values (the halo request's VALU burst, once per chunk) from spilled
            # SGPRs: a dozen v_readlane in 288 MFMAs' worth of loop, no v_writelane, nothing in scratch
            assert len(inside) <= 12 and all(i.startswith("v_readlane") for i in inside), (key, inside[:5])
        else:
            assert not inside, (key, inside[:5])


def test_one_chunk_bf16_block_instances_keep_their_registers(code_object):
    """DESIGN.md 3.7 "Round 4": the one-chunk form of the bf16 block kernel exists to get the shortcut's accumulators and
    the staging registers out of the unrolled steps -- every `block_bf16_one_kernel` / `block_bf16_two_kernel` instance fits its 256 registers with
    NOTHING in scratch (detector.layer.1's 22 spilled registers were 0.40 -> 0.33 ms), and layer1's keeps four pixel
    blocks per weight fragment (MB = 4) at that."""
    funcs, meta = code_object
    inst = {}
    for name in funcs:
        m = re.match(r"_ZN3fpc21block_bf16_(?:one|two)_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEEvNS_11BlockBfArgsE$", name)
        if m:
            inst[tuple(int(v) for v in m.groups())] = name
    assert len(inst) >= 6, sorted(inst)      # (five one-chunk instances and layer_in.1's two-chunk one)
    for key, name in inst.items():
        m = meta[name]
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (key, m)
        assert not any(i.startswith("scratch_") for _, i in funcs[name]), key
    layer1 = [k for k in inst if k[4] == 64 and k[2] == 1]          # (TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP): KC = 64, stride 1
    assert layer1 and all(k[7] == 4 for k in layer1), layer1


def test_round5_stem_has_no_valu_instruction_between_its_mfmas(code_object):
    """DESIGN.md 3.2 (round 5): stem_pool2_kernel exists because a VALU instruction costs fp32 MFMA time on gfx950 -- the
    shipped instances hold NO VALU instruction between their first and last-but-one MFMA group (per-channel LDS bases, the
    bias as a K step), name no AGPR (accumulators straight into ds_write), fit 128 registers (four workgroups per CU)
    with nothing in scratch, and issue well under half of stem_pool_kernel's VALU instructions."""
    funcs, meta = code_object
    new = {n: b for n, b in funcs.items() if re.match(r"_ZN3fpc17stem_pool2_kernelILi[13]EEEvNS_12StemPoolArgsE$", n)}
    old = {n: b for n, b in funcs.items() if re.match(r"_ZN3fpc16stem_pool_kernelILi[13]EEEvNS_12StemPoolArgsE$", n)}
    assert len(new) == 2 and len(old) == 2, (sorted(new), sorted(old))
    valu = lambda body: sum(1 for _, i in body if i.startswith("v_") and not i.startswith("v_mfma"))
    for name, body in new.items():
        m = meta[name]
        assert m["agpr_count"] == 0 and m["vgpr_count"] <= 128, (name, m)
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
        seq = [i for _, i in body]
        mf = [k for k, i in enumerate(seq) if i.startswith("v_mfma")]
        assert {seq[k].split()[0] for k in mf} == {"v_mfma_f32_32x32x2_f32"}
        cin = 3 if "ILi3E" in name else 1
        assert len(mf) == 4 * (cin * 25 + 1), (name, len(mf))            # 25 steps per channel + the bias step, 2 x 2 blocks
        # (the compiler sinks the last step's MFMAs of the SECOND half under the first half's epilogue -- its address arithmetic
        # and pooling: everything in front of the last four steps' sixteen MFMAs is the K loop proper)
        inside = [i for i in seq[mf[0]:mf[-17]] if i.startswith("v_") and not i.startswith("v_mfma")]
        assert not inside, (name, inside[:6])
    for cin in (1, 3):
        n_new = valu(next(b for n, b in new.items() if "ILi%dE" % cin in n))
        n_old = valu(next(b for n, b in old.items() if "ILi%dE" % cin in n))
        assert 5 * n_new < 2 * n_old, (cin, n_new, n_old)      # (972 -> 365 and 839 -> 319 when written, the partial-tile path included)


def test_round5_bf16_conv_transpose_kernel_keeps_its_registers(code_object):
    """convt_bf16_kernel (csrc/convt_bf16.h): 128 accumulators (four output parities x two pixel blocks) + the nine-fragment ring
    + the pixel ring fit the 256 registers of two waves per SIMD with nothing in scratch, and a chunk is 4 x 18 MFMAs."""
    funcs, meta = code_object
    inst = [n for n in funcs if n.startswith("_ZN3fpc17convt_bf16_kernelI")]
    assert len(inst) == 1, inst
    m = meta[inst[0]]
    assert m["vgpr_count"] <= 256 and m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, m
    body = funcs[inst[0]]
    assert not any(i.startswith("scratch_") for _, i in body)
    loop = _chunk_loop(body)
    assert loop is not None and loop[2] == 72, loop


def test_round5_conv2_kernel_is_lean_and_spill_free(code_object):
    """conv2_mfma_kernel (csrc/conv_mfma.h): the two default-plan instances hold nothing in scratch, no AGPR, and well under
    half of the VALU instructions of the conv_mfma_kernel instances they replace (1 372 / 1 255 -> 473 / 436 when written)."""
    funcs, meta = code_object
    valu = lambda body: sum(1 for _, i in body if i.startswith("v_") and not i.startswith("v_mfma"))
    pairs = 0
    for name, body in funcs.items():
        m = re.match(r"_ZN3fpc17conv2_mfma_kernelI(.+)EEvNS_8ConvArgsE$", name)
        if not m:
            continue
        old = "_ZN3fpc16conv_mfma_kernelI%sEEvNS_8ConvArgsE" % m.group(1)
        assert old in funcs, old
        mm = meta[name]
        assert mm["agpr_count"] == 0 and mm["vgpr_spill_count"] == 0 and mm["private_segment_fixed_size"] == 0, (name, mm)
        assert 2 * valu(body) < valu(funcs[old]), (name, valu(body), valu(funcs[old]))
        pairs += 1
    assert pairs == 2, pairs
