import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d["value"]), round(d["ms_per_step"],4), {k:round(v["ms_total"]/v["launches"]*1e3,1) for k,v in d["kernel_times_ms"].items()})
