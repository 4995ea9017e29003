"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile) per kernel instantiation:
    hipcc ... -Rpass-analysis=kernel-resource-usage -c sweep.hip 2> build.log ; python tools/resource_report.py build.log [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"Function Name: ", txt)[1:]
print(f"{'kernel':58s} VGPR AGPR scratch occ spillV LDS")
for b in blocks:
    name = b.split()[0]
    if flt and flt not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    m = re.search(r"(sweep_\w+?_kernel)I(.*?)EEv", name)
    tag = (m.group(1) + "<" + m.group(2).replace("ELb", ",").replace("ELi", ",").replace("Li", "").replace("Lb", "") + ">") if m else name[:58]
    print(f"{tag:58s} {g('    VGPRs'):4d} {g('AGPRs'):4d} {g('ScratchSize .bytes/lane.'):7d} {g('Occupancy .waves/SIMD.'):3d} {g('VGPRs Spill'):6d} {g('LDS Size .bytes/block.')}")
