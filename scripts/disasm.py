"""Disassemble the gfx950 code object inside a hipcc object file.

    python scripts/disasm.py nafae_amd/csrc/wino.o [out.s]

Prints per kernel: instruction count, MFMA count, scratch (spill) instructions, and writes the full listing."""
import os
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, os.path.basename(obj))
        shutil.copy(obj, o)
        subprocess.check_call([OBJDUMP, "--offloading", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        return subprocess.check_output([OBJDUMP, "-d", os.path.join(d, co[0])], text=True)


def main():
    obj = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/" + os.path.basename(obj) + ".s"
    asm = disassemble(obj)
    open(out, "w").write(asm)
    name, n, mfma, scr = None, 0, 0, 0
    for line in asm.splitlines() + ["<end>:"]:
        if line.endswith(">:") and "<" in line:
            if name:
                print("%-90s %6d instr %5d mfma %4d scratch" % (name[:90], n, mfma, scr))
            name, n, mfma, scr = line[line.index("<") + 1:-2], 0, 0, 0
        elif name and line.strip():
            n += 1
            mfma += "v_mfma" in line
            scr += "scratch_" in line
    print("listing:", out)


if __name__ == "__main__":
    main()
