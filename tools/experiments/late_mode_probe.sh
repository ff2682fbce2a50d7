#!/bin/bash
# the late phase of one chain is bimodal between runs (0.8 or 1.6 ms on the critical path): which part of it?  per run: the medians of the last
# stage's strand times and of the stages before it (VPBS_TRACE_WITNESS lines), next to the tool's own split
for rep in 1 2 3 4 5 6; do
  VPBS_TRACE_POOLS=1 VPBS_TRACE_WITNESS=1 VPBS_IVC_CHAINS=1 python tools/prove_ivc.py 1024 728 16 150 2>/tmp/err.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_split']; print('run $rep:', round(d['ms_per_step'],3), 'ms/step; late', round(s['witness_late_phase_host'],3), 'prove', round(s['prove_step'],3))"
  python - <<'PY'
import re, statistics as st
lines = open('/tmp/err.txt', errors='replace').read().splitlines()
strand = [(float(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))) for l in lines for m in [re.search(r'stage 6, \d+ threads: \d+ generators in strands ([\d.]+) ms \(last thread started after ([\d.]+) ms; strands took ([\d.]+) \.\. ([\d.]+) ms\)', l)] if m]
wide = [float(m.group(2)) + float(m.group(1)) for l in lines for m in [re.search(r'14 threads: 39 narrow levels ([\d.]+) ms, 5 wide levels ([\d.]+) ms', l)] if m]
st1 = [float(m.group(1)) + float(m.group(2)) for l in lines for m in [re.search(r'14 threads: 150 narrow levels ([\d.]+) ms, 12 wide levels ([\d.]+) ms', l)] if m]
if strand:
    print('   stage 6: total %.2f, last start %.2f, shortest %.2f, longest %.2f | stage 5 levels %.2f | stage 1 levels %.2f (medians over %d steps)' % (
        st.median(x[0] for x in strand), st.median(x[1] for x in strand), st.median(x[2] for x in strand), st.median(x[3] for x in strand), st.median(wide) if wide else -1, st.median(st1) if st1 else -1, len(strand)))
print('   ' + ' | '.join(l[7:60] for l in lines if l.startswith('[pool]')))
PY
done
