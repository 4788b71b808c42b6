L=$1; O=$2; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pp; timeout 60 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -o p -- python $R/tools/one_kernel.py $L $O 512 5 > /tmp/pp.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("igemm", "wgrad", "bwd_pair")):
        continue
    acc[n[:n.index("(")].replace("(anonymous namespace)::", "").replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
if not f: print(open("/tmp/pp.log").read()[-1500:])
PY
done
