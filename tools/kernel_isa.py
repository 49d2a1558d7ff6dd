#!/usr/bin/env python
"""Per-kernel hashes of the gfx950 machine code in the library's objects, to show that an edit of
the host code (a pruned switch, a moved function) left every kernel that is still there bit for bit
what it was -- the check to make when no GPU is at hand to re-run the parity tests.

    python tools/kernel_isa.py snapshot OUT.json      # hash every kernel of boxtree_amd/csrc/*.o
    python tools/kernel_isa.py diff OLD.json NEW.json  # removed / added / CHANGED kernels

A kernel's hash covers its disassembled instructions (llvm-objdump -d of the code object
llvm-objdump --offloading extracts), addresses and encodings dropped, and its kernel descriptor
(registers, LDS, scratch: `.amdhsa_` directives are not in the disassembly, so the 64 descriptor bytes
`<kernel>.kd` are hashed instead, less the code's offset).
"""

from __future__ import annotations

import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj, work):
    base = os.path.join(work, os.path.basename(obj))
    shutil.copy(obj, base)
    subprocess.run([f"{BIN}/llvm-objdump", "--offloading", base], check=True, capture_output=True)
    cos = [f for f in glob.glob(base + ".*") if "gfx950" in f]
    out = {}
    for co in cos:
        dis = subprocess.run([f"{BIN}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True,
                             capture_output=True, text=True).stdout
        name, body = None, []
        blocks = {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if name is not None:
                    blocks[name] = body
                name, body = m.group(1), []
                continue
            if name is None or not line.strip():
                continue
            # "\ts_load_dwordx2 s[0:1], s[4:5], 0x0      // 000000001000: ..." -> the instruction
            ins = line.split("//")[0].strip()
            ins = re.sub(r"^[0-9a-f]+:\s*", "", ins)
            if ins:
                body.append(ins)
        if name is not None:
            blocks[name] = body
        # kernel descriptors (.rodata): symbol table gives offset + size
        syms = subprocess.run([f"{BIN}/llvm-readelf", "-sW", co], check=True, capture_output=True,
                              text=True).stdout
        secs = subprocess.run([f"{BIN}/llvm-readelf", "-SW", co], check=True, capture_output=True,
                              text=True).stdout
        ro = re.search(r"\]\s+\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
        data = open(co, "rb").read()
        kds = {}
        if ro:
            addr, off = int(ro.group(1), 16), int(ro.group(2), 16)
            for ln in syms.splitlines():
                f = ln.split()
                if len(f) >= 8 and f[-1].endswith(".kd"):
                    a, size = int(f[1], 16), int(f[2])
                    kd = bytearray(data[off + a - addr: off + a - addr + size])
                    kd[16:24] = b"\0" * 8       # KERNEL_CODE_ENTRY_BYTE_OFFSET: where the code sits
                    kds[f[-1][:-3]] = bytes(kd)
        for k, body in blocks.items():
            h = hashlib.sha256("\n".join(body).encode())
            h.update(kds.get(k, b""))
            out[k] = {"hash": h.hexdigest()[:16], "instructions": len(body), "kernel": k in kds}
    return out


def snapshot(path):
    work = tempfile.mkdtemp(prefix="kisa_")
    res = {}
    try:
        for obj in sorted(glob.glob(os.path.join(ROOT, "boxtree_amd", "csrc", "*.o"))):
            res[os.path.basename(obj)] = kernels_of(obj, work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    with open(path, "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    n = sum(1 for o in res.values() for k in o.values() if k["kernel"])
    print(f"{path}: {n} kernels in {len(res)} objects, "
          f"{sum(k['instructions'] for o in res.values() for k in o.values())} instructions")


def demangle(names):
    if not names:
        return {}
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines())) if p.returncode == 0 else {n: n for n in names}


def diff(a, b):
    old, new = json.load(open(a)), json.load(open(b))
    removed, added, changed, same = [], [], [], 0
    for obj in sorted(set(old) | set(new)):
        o, n = old.get(obj, {}), new.get(obj, {})
        for k in sorted(set(o) | set(n)):
            if k not in n:
                removed.append((obj, k, o[k]["instructions"]))
            elif k not in o:
                added.append((obj, k, n[k]["instructions"]))
            elif o[k]["hash"] != n[k]["hash"]:
                changed.append((obj, k, o[k]["instructions"], n[k]["instructions"]))
            else:
                same += 1
    dm = demangle([k for _, k, *_ in removed + added + changed])
    print(f"identical: {same}   removed: {len(removed)}   added: {len(added)}   CHANGED: {len(changed)}")
    for tag, rows in (("removed", removed), ("added", added), ("CHANGED", changed)):
        for r in rows:
            print(f"  {tag:8s} {r[0]:18s} {dm.get(r[1], r[1])[:150]}  [{' -> '.join(map(str, r[2:]))} instr]")
    return 1 if changed else 0


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "snapshot":
        snapshot(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "diff":
        sys.exit(diff(sys.argv[2], sys.argv[3]))
    else:
        sys.exit(__doc__)
