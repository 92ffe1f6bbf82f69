"""Per-kernel register / scratch / spill figures from an `hipcc -S --cuda-device-only` listing:  python tools/isa_stats.py file.s [filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt not in name:
        continue
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):4d} sgpr {g('sgpr_count'):4d} scratch {g('private_segment_fixed_size'):5d} vgpr_spill {g('vgpr_spill_count'):4d} lds {g('group_segment_fixed_size')}")
