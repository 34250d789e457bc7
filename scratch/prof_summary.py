import csv,glob,re,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_stats.csv")[0]
nsteps=float(sys.argv[2])
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
ncalls=sum(int(r["Calls"]) for r in rows)
grp={}
def add(k,t): grp[k]=grp.get(k,0)+t
for r in rows:
    n=r["Name"]; t=float(r["TotalDurationNs"])
    if n.startswith("Cijk"): add("rocblas",t)
    elif "batch_norm" in n: add("torch_bn",t)
    elif "at::native" in n: add("torch_other",t)
    elif "crf::bn_" in n: add("crf_bn",t)
    elif "crf::wgrad" in n: add("crf_wgrad",t)
    elif "crf::" in n and any(k in n for k in ("stats_kernel","forward_kernel","bwd_reduce","bwd_params","bwd_input","bwd_dump","moments","reduce_partials")): add("crf_pointconv",t)
    elif "crf::" in n and any(k in n for k in ("step","sim_","bwd_edge","bwd_scatter")): add("crf_meanfield",t)
    elif "crf::" in n or "rocprim" in n: add("crf_other",t)
    else: add("other",t)
print("total kernel ms per step %.2f ; launches per step %d" % (tot/nsteps/1e6, ncalls/nsteps))
for k,v in sorted(grp.items(), key=lambda kv:-kv[1]): print("  %-14s %.2f ms/step" % (k, v/nsteps/1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 30]:
    print("%-72s calls %5s tot %7.2f ms avg %8.1f us" % (re.sub(r"\(.*","",r["Name"]).replace("void ","")[:72], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
