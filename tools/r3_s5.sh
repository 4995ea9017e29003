cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s5; rm -rf $O; mkdir -p $O
cd $R
for w in dr fibre; do timeout 100 python tools/first_call.py $w >> $O/first_call.txt 2>&1; done
timeout 100 python tools/first_call.py dr 3.0 >> $O/first_call.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --hip-trace --kernel-trace --stats -d $O/prof_first -o x -- python $R/tools/first_call.py dr > $O/first.log 2>&1
ls -R $O/prof_first | head -20
python - <<PY
import sqlite3, glob
for db in glob.glob("$O/prof_first/**/x_results.db", recursive=True):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    print([n for n in names if 'top' in n or 'hip' in n.lower() or 'api' in n.lower()][:40])
    for v in ('top',):
        try:
            for row in c.execute("select * from top limit 30"): print(row)
        except Exception as e: print('ERR', e)
PY
rm -rf $O/prof_first
cat $O/first_call.txt
