# sample clocks / power while the learner probe runs (is the fp32-MFMA learner power-limited?)
(for i in $(seq 1 200); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.4; done) > /tmp/smi.log &
SPID=$!
python tools/learner_probe.py > /tmp/lp.log 2>&1
python tools/learner_probe.py >> /tmp/lp.log 2>&1
kill $SPID
grep tile /tmp/lp.log | tail -n 5
sort -t: -k5 -n /tmp/smi.log | awk -F'Power \\(W\\): ' '{print $2+0, $0}' | sort -n | tail -n 6 | cut -c1-200
