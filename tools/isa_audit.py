"""Serialised loads inside loops, read off the ISA (no GPU needed): python tools/isa_audit.py [file.hip ...] [-D...]

For every kernel of the given sources (default: all of myria3d_amd/csrc/*.hip) the device code is compiled to assembly
(hipcc --cuda-device-only -S, gfx950) and every basic block that LLVM marks as part of a loop is scanned for
``s_waitcnt vmcnt(0)`` with at most two vector-memory loads issued since the previous full wait: a dependent round trip per
iteration.  That is the pattern behind round 4's last finding ("serialised loads in the simple kernels":
loads under a null check or an ``if (act)`` inside a row loop are basic blocks of their own, each with its own wait, and
``#pragma unroll`` only repeats them).  Prints, per kernel: loop blocks, flagged blocks, loads per full wait in the worst
block.  A flagged block is a hint, not a verdict: a loop that runs once per wave, or one whose wave count hides the
latency (the LFA forward at 8 waves per SIMD), costs nothing."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)")
FULL = re.compile(r"^\s*s_waitcnt .*vmcnt\(0\)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):(.*)$")
KERNEL = re.compile(r"^(_Z\w+|[A-Za-z_]\w*):\s*;\s*@")


def audit(path, defines):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-variable", "-w",
               "--cuda-device-only", "-S", path, "-o", tmp.name] + defines
        subprocess.run(cmd, check=True, cwd=os.path.dirname(path), stderr=subprocess.DEVNULL)
        text = open(tmp.name).read().split("\n")
    rows, kernel, in_loop, since, blocks, flagged, worst = [], None, False, 0, 0, 0, None

    def close():
        if kernel is not None and blocks:
            rows.append((flagged, blocks, worst if worst is not None else -1, kernel))

    for line in text:
        m = KERNEL.match(line)
        if m:
            close()
            kernel, in_loop, since, blocks, flagged, worst = m.group(1), False, 0, 0, 0, None
            continue
        if kernel is None:
            continue
        m = LABEL.match(line)
        if m:
            in_loop = "Loop" in m.group(2)
            blocks += in_loop
            since = 0
            continue
        if ".Lfunc_end" in line:
            close()
            kernel = None
            continue
        if not in_loop:
            continue
        if LOAD.match(line):
            since += 1
        elif FULL.match(line):
            if 0 < since <= 2:
                flagged += 1
            if since and (worst is None or since < worst):
                worst = since
            since = 0
    close()
    return rows


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), text=True, capture_output=True)
        return out.stdout.split("\n")
    except OSError:
        return names


if __name__ == "__main__":
    defines = [a for a in sys.argv[1:] if a.startswith("-D")]
    files = [os.path.abspath(a) for a in sys.argv[1:] if not a.startswith("-D")] or sorted(glob.glob(os.path.join(ROOT, "myria3d_amd", "csrc", "*.hip")))
    for f in files:
        rows = [r for r in audit(f, defines) if r[0]]
        print(f"== {os.path.basename(f)}: {len(rows)} kernel(s) with a full wait behind <= 2 loads inside a loop")
        names = demangle([r[3] for r in rows])
        for (fl, bl, wo, _), name in sorted(zip(rows, names), reverse=True)[:40]:
            print(f"   {fl:3d} flagged of {bl:3d} loop blocks, fewest loads per full wait {wo:2d}   {name[:110]}")
