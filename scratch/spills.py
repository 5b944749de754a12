"""List SGPR/VGPR spill counts per kernel from a -save-temps .s file (usage: spills.py file.s)."""
import re, sys
txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
    print(f"sgpr {g('sgpr_count'):4d} spill {g('sgpr_spill_count'):4d} | vgpr {g('vgpr_count'):4d} spill {g('vgpr_spill_count'):4d} | {name[:90]}")
