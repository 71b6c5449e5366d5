import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "%.3fM"%(d["value"]/1e6), "ms/step %.4f"%d["ms_per_step"], "kernel %.4f"%d["roofline"]["kernel_ms"], d["state_mix"]["mean_contacts"])
