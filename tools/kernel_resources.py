"""Register / LDS / spill table of every kernel in csrc/sf_api.hip (hipcc -Rpass-analysis=kernel-resource-usage), optionally
against another git revision: `python tools/kernel_resources.py [--against REV] [--md out.md]`.  CPU only (cross-compiles).
Round 4: a dead branch added to the GEMM epilogues cost the 128-VGPR variants 20 spilled registers -- invisible without this."""
import argparse
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("VGPRs:", "AGPRs:", "VGPRs Spill:", "SGPRs Spill:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "LDS Size [bytes/block]:")


def analyse(src_root):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared",
               "-Wno-comment", "-I" + os.path.join(src_root, "include"), "-Rpass-analysis=kernel-resource-usage",
               os.path.join(src_root, "slowfast_amd", "csrc", "sf_api.hip"), "-o", os.path.join(tmp, "x.so")]
        err = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp).stderr
    out, name = {}, None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            out[name] = {}
        for k in KEYS:
            if name and k in line:
                out[name][k] = int(line.split(k)[1].split()[0])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--against", default=None, help="git revision to compare with")
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    new = analyse(ROOT)
    old = None
    if a.against:
        with tempfile.TemporaryDirectory() as tmp:
            tar = subprocess.run(["git", "-C", ROOT, "archive", a.against, "slowfast_amd/csrc", "include"], capture_output=True).stdout
            subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
            old = analyse(tmp)
    lines = ["| kernel | VGPRs | AGPRs | spilled VGPRs | scratch B/lane | waves/SIMD | LDS B |" + (" change vs %s |" % a.against if old else ""),
             "|---|---:|---:|---:|---:|---:|---:|" + ("---|" if old else "")]
    for k, v in sorted(new.items()):
        row = "| `%s` | %d | %d | %d | %d | %d | %d |" % (k[:110], v.get(KEYS[0], 0), v.get(KEYS[1], 0), v.get(KEYS[2], 0),
                                                        v.get(KEYS[4], 0), v.get(KEYS[5], 0), v.get(KEYS[6], 0))
        if old is not None:
            o = old.get(k)
            row += (" new |" if o is None else (" |" if o == v else " VGPRs %d -> %d, spills %d -> %d, waves %d -> %d |" % (
                o.get(KEYS[0], 0), v.get(KEYS[0], 0), o.get(KEYS[2], 0), v.get(KEYS[2], 0), o.get(KEYS[5], 0), v.get(KEYS[5], 0))))
        lines.append(row)
    text = "\n".join(lines)
    if a.md:
        open(a.md, "w").write("# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n\n" + text + "\n")
    print(text)


if __name__ == "__main__":
    main()
