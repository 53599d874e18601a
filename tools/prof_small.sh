cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profB
rocprofv3 --kernel-trace --stats -d /tmp/profB -- python $GRAFT_REPO_ROOT/tools/launch_bound.py $1 $2 > /tmp/profB.log 2>&1
python - <<'PY'
import sqlite3,glob
db=glob.glob("/tmp/profB/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
rows=list(cur.execute("select name,total_calls,total_duration from top_kernels"))
tot=sum(r[2] for r in rows if "rgm::" in r[0])
calls=sum(r[1] for r in rows if "rgm::" in r[0])
print("rgm kernels: calls",calls,"total ms",tot/1e6)
PY
grep -E "B=|graph" /tmp/profB.log
