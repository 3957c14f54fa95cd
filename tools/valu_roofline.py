#!/usr/bin/env python3
"""Derives the integer-VALU roofline of the Skein-512 kernel and writes profiles/valu_roofline.json (bench.py reads it).

  peak = SIMDs x bytes per wavefront-call / sum over the kernel's VALU opcodes of (count per Threefish call x measured issue cost)

  counts  from the ISA hipcc emits for cw::skein_slice_kernel<8, true> (csrc/skein_kernels.hip; two Threefish bodies per loop)
  costs   from tools/ubench.hip run on the MI355X: profiles/r02_ubench_valu_issue.txt (ns per wavefront-instruction per SIMD
          at 8 wavefronts per SIMD, whole chip, wall clock)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UBENCH = os.path.join(ROOT, "profiles", "r02_ubench_valu_issue.txt")
OPS = {"v_xor_b32": "v_xor_b32", "v_alignbit_b32": "v_alignbit_b32", "v_lshl_add_u64": "u64 add (v_lshl_add_u64)"}


def costs():
    out = {}
    for line in open(UBENCH):
        m = re.match(r"(.+?)\s+waves/SIMD=8 .*\(ns per instr ([0-9.]+)\)", line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def counts():
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "skein.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "compute_war_amd", "csrc", "skein_kernels.hip"), "-o", s], check=True, capture_output=True)
        text = open(s).read()
    m = re.search(r"^(_ZN2cw18skein_slice_kernelILi8ELb1EEE\w*):.*?s_endpgm", text, re.S | re.M)
    body = m.group(0)
    c = {op: len(re.findall(r"^\s+" + op + r"(?:_e32|_e64)?\s", body, re.M)) for op in OPS}
    total_valu = len(re.findall(r"^\s+v_\w+", body, re.M))
    return c, total_valu, m.group(1)


def main():
    cost, (cnt, total_valu, sym) = costs(), counts()
    bodies = 2   # one Threefish-512 call per 64-byte half of a 128-byte line, two per loop iteration
    per_call = {op: cnt[op] / bodies for op in OPS}
    ns = sum(per_call[op] * cost[OPS[op]] for op in OPS)
    simds, bytes_per_call = 256 * 4, 64 * 64   # 64 lanes x 64 message bytes
    peak = simds * bytes_per_call / ns          # bytes per ns = GB/s
    out = {
        "kernel": "cw::skein_slice_kernel<8, true>", "symbol": sym,
        "instr_per_threefish_call": per_call, "all_valu_in_kernel": total_valu,
        "ns_per_wave_instr_per_simd": {op: cost[OPS[op]] for op in OPS},
        "ns_per_wave_call_per_simd": round(ns, 1), "simds": simds, "bytes_per_wave_call": bytes_per_call,
        "peak_GBps": round(peak, 1),
        "sources": ["profiles/r02_ubench_valu_issue.txt (tools/ubench.hip on the MI355X)", "hipcc -S of compute_war_amd/csrc/skein_kernels.hip"],
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_roofline.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
