#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command -> text summary under gpurun_out/ (copy into profiles/ to keep it)
#   [PROF_OUT=dir] tools/prof_run.sh <name> <command...>
export TMPDIR=/tmp
R=$PWD
name=$1; shift
rm -rf /tmp/prof_$name
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- "$@" > /tmp/prof_$name.log 2>&1)
DB=$(find /tmp/prof_$name -name "*.db" | head -1)
OUTD=${PROF_OUT:-$R/gpurun_out/r5}
mkdir -p $OUTD
{ echo "# rocprofv3 --kernel-trace --stats -- $*"; [ -n "$SFMI_COMMIT" ] && echo "# taken at commit $SFMI_COMMIT"; grep -a '^{"metric"' /tmp/prof_$name.log | python3 -c "import sys,json; [print(json.dumps({k:v for k,v in json.loads(l).items() if k in (\"metric\",\"value\",\"ms_per_step\",\"steps\",\"roofline\",\"ar_loop\",\"stages_ms\")})) for l in sys.stdin]"; python $R/tools/prof_summary.py $DB 40; } > $OUTD/prof_$name.txt
if [ -z "$DB" ]; then echo "no rocpd database produced; log tail:"; tail -n 20 /tmp/prof_$name.log; fi
tail -n 45 $OUTD/prof_$name.txt
