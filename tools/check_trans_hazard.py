#!/usr/bin/env python3
"""Scan the gfx950 code objects inside libinferix_hip.so for the hazard DESIGN 9 records: a VALU instruction that reads the
result of a transcendental (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos, f32 and f16 forms) in the very next
instruction slot.  The compiler pads its own instructions (s_nop or an independent instruction in between); inline asm is
invisible to its hazard recogniser, so a recompile can silently put an asm consumer directly behind the transcendental.
Exit status 1 and one line per finding if any kernel has such a pair.

usage: tools/check_trans_hazard.py [path/to/libinferix_hip.so]
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_(f32|f16|bf16|legacy_f32|iflag_f32)")
VALU = re.compile(r"^v_(?!readlane|readfirstlane|writelane|nop)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def code_objects(so_path: str, outdir: str):
    """The gfx950 ELF images of every clang offload bundle in the library's .hip_fatbin section."""
    fat = os.path.join(outdir, "fatbin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so_path, fat], check=True)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    for m in re.finditer(magic, data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(magic))
        p = base + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(outdir, f"co_{len(out)}.elf")
                open(path, "wb").write(data[base + off:base + off + size])
                out.append(path)
    return out


def regs(operand_text: str):
    s = set()
    for a, lo, hi in REG.findall(operand_text):
        if a:
            s.add(int(a))
        else:
            s.update(range(int(lo), int(hi) + 1))
    return s


def scan(elf: str):
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", elf], capture_output=True, text=True, check=True).stdout
    findings = []
    kernel, prev = "?", None
    for line in txt.splitlines():
        line = line.strip()
        if line.endswith(">:"):
            kernel, prev = line.split("<")[-1][:-2], None
            continue
        if not line or line.startswith(("/", ";", ".")) or ":" in line.split()[0]:
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if prev is not None and VALU.match(op) and not op.startswith("v_mfma"):
            # operands after the first one are sources (VOP: dst, src0, src1, ...); "+v" asm operands appear as dst AND src
            parts = ins[len(op):].split(",")
            srcs = regs(",".join(parts[1:]))
            if prev[1] & srcs:
                findings.append((kernel, prev[0], ins))
        prev = None
        if TRANS.match(op):
            dst = regs(ins[len(op):].split(",")[0])
            prev = (ins, dst)
    return findings


def main() -> int:
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "inferix_amd", "libinferix_hip.so")
    bad = []
    with tempfile.TemporaryDirectory() as td:
        objs = code_objects(so, td)
        if not objs:
            print("no gfx950 code object found in", so)
            return 2
        for o in objs:
            bad += scan(o)
    for k, a, b in bad:
        print(f"{k}: `{a}` is read by the next instruction `{b}`")
    print(f"{len(objs)} code objects scanned, {len(bad)} transcendental -> VALU pairs without a wait state")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
