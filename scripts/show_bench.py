import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("backward"))
print(d.get("cpu_baseline"))
for e in d.get("extra",[]):
    print(round(e["ms"],3), "|", e["config"][:95], "|", e.get("schedule","")[:70], "|", e.get("roofline",{}).get("frac"), e.get("roofline",{}).get("traffic"), e.get("cpu_epoch",{}).get("ms"))
