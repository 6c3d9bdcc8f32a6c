import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); print("%.4e ms/step %.4f enqueue %.4f kern region %.4f isolated %.4f" % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_isolated_mean"]))
