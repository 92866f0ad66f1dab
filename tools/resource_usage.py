"""Summarise `make asm`'s -Rpass-analysis=kernel-resource-usage output: one line per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/resource_usage.txt").read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"^void ", "", re.sub(r"\(.*", "", dn))
    print("{:58s} vgpr {:>4} agpr {:>3} sgpr {:>3} scratch {:>5} occ {} lds {}".format(
        dn, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
