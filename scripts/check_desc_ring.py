#!/usr/bin/env python3
"""Verifies, on the generated gfx950 code of kernels_desc.hip, the contract of k_describe's register ring (S3D_DX_PF):
inside the march loop and its prologue no instruction OUTSIDE the inline-asm blocks reads or writes a ring register (v112-v123),
the kernel has no scratch (scratch traffic counts in vmcnt like the ring's loads) and no SGPR spill instruction (v_readlane /
v_writelane) sits inside the march loop.  CPU box; exit code 1 on a violation.

    python3 scripts/check_desc_ring.py [-DS3D_...]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3dsift_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()


def regs_of(line):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "kernels_desc.hip")],
                           capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-3000:])
        text = open(out).read().split("\n")
    bad = 0
    kernels = [i for i, l in enumerate(text) if re.match(r"^_ZN3s3d10k_describe.*:\s", l)]
    for ki in kernels:
        name = text[ki].split(":")[0]
        end = next(i for i in range(ki, len(text)) if "s_endpgm" in text[i])
        body = text[ki:end]
        ring_loads = [i for i, l in enumerate(body) if re.search(r"global_load_dwordx4 v\[1(12|20|24):", l)]
        if not ring_loads:
            print(f"{name[:60]}: no ring (S3D_DX_PF off?)")
            continue
        ring = set()
        for i in ring_loads:
            ring |= regs_of(body[i].split(",")[0])
        for i, l in enumerate(body):
            if re.search(r"global_load_dwordx2 v\[1(16|18|10):", l):
                ring |= regs_of(l.split(",")[0])
        # blocks of the march loop: every label whose loop annotation names the header of the innermost loop around the first turn
        # (the turns carry a counted wait in front of their loads), plus the prologue in front of that loop
        turn = next(i for i in ring_loads if any("s_waitcnt vmcnt(" in body[j] and "vmcnt(0)" not in body[j] for j in range(i - 12, i)))
        hdr = None
        for i in range(turn, 0, -1):
            m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(?:in Loop: Header=(BB\d+_\d+)|=>\s*This Loop Header|=>This Loop Header)", body[i])
            if m:
                hdr = m.group(2) or m.group(1)[2:]
                break
        assert hdr, "march loop header not found"
        marks = [False] * len(body)
        i = 0
        while i < len(body):
            if re.match(r"^\.LBB\d+_\d+:", body[i]):
                j = i + 1
                note = body[i]
                while j < len(body) and body[j].strip().startswith(";"):
                    note += body[j]; j += 1
                inside = hdr in note
                k = j
                while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]):
                    marks[k] = inside; k += 1
                i = k
            else:
                i += 1
        for i in range(ring_loads[0], turn):  # prologue -> loop
            marks[i] = True
        in_asm = False
        viol = []
        for i, l in enumerate(body):
            if "#ASMSTART" in l: in_asm = True; continue
            if "#ASMEND" in l: in_asm = False; continue
            if not marks[i] or in_asm or l.strip().startswith(";") or l.strip().startswith("."): continue
            if regs_of(l.split(";")[0]) & ring:
                viol.append((i, l.strip()))
        scratch = [l.strip() for l in body if "scratch_" in l]
        # SGPR spills (v_writelane / v_readlane) are tolerated in the per-keypoint set-up and epilogue, never inside the march loop
        loop_spills = [l.strip() for i, l in enumerate(body) if marks[i] and i >= turn - 40 and re.search(r"v_(readlane|writelane)_b32", l)]
        all_spills = sum(1 for l in body if re.search(r"v_(readlane|writelane)_b32", l))
        print(f"{name[:60]}: ring registers {sorted(ring)[0]}..{sorted(ring)[-1]} ({len(ring)}), {len(ring_loads)} ring loads, "
              f"{len(viol)} outside accesses, {len(scratch)} scratch instructions, SGPR spill instructions: {len(loop_spills)} in the march loop / {all_spills} in the kernel")
        bad += len(loop_spills)
        for i, l in viol[:10]:
            print(f"   line +{i}: {l}")
        bad += len(viol) + len(scratch)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
