"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (VGPR / SGPR / scratch / occupancy / LDS per kernel)."""
import re
import subprocess
import sys


def main(path, only_f32=True):
    txt = open(path).read()
    blocks = txt.split("Function Name: ")[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    seen = set()
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        d = re.sub(r"\(mgacbam::\w+\)", "", d).replace("mgacbam::", "").replace("void ", "")[:64]
        if only_f32 and ("__half" in d or "bfloat" in d):
            continue
        if d in seen:
            continue
        seen.add(d)
        print("%-66s vgpr=%4s agpr=%3s sgpr=%4s scratch=%4s occ=%s lds=%s" % (
            d, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
            g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main(sys.argv[1], only_f32="--all" not in sys.argv)
