import json,sys
r=json.load(open("gpurun_out/bwd_arbiter.json"))
key=sys.argv[1] if len(sys.argv)>1 else sorted(r)[0]
print(list(r)); rows=r[key]
v=sorted(rows.items(), key=lambda kv:-kv[1]["rms_ratio"])
import statistics
print(len(rows), "median", statistics.median(x["rms_ratio"] for x in rows.values()), "n>1.5:", sum(x["rms_ratio"]>1.5 for x in rows.values()))
for k,x in v[:40]: print("%-55s rms %8.2f p99 %8.2f  hip %.2e torch %.2e n=%d"%(k,x["rms_ratio"],x["p99_ratio"],x["hip_rms_err_over_rms"],x["torch_fp32_rms_err_over_rms"],x["numel"]))
