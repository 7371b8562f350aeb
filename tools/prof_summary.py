"""Summarises rocprofv3 sqlite outputs (kernel trace + PMC passes) for the render kernel into text."""
import glob, os, sqlite3, sys

def main(root):
  for db in sorted(glob.glob(os.path.join(root, '**', '*_results.db'), recursive=True)):
    con = sqlite3.connect(db)
    cur = con.cursor()
    print(f'== {os.path.relpath(db, root)}')
    try:
      for name, calls, tot, avg, pct in cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 4'):
        print(f'  kernel {name[:90]:90s} calls={calls} total_ms={tot/1e3:.3f} avg_ms={avg/1e3:.4f} pct={pct:.1f}')   # top_kernels durations are microseconds
    except Exception as e:
      print('  (no top_kernels)', e)
    try:
      rows = cur.execute("select kernel_name, counter_name, sum(value), count(*), avg(duration), max(vgpr_count), max(accum_vgpr_count), max(scratch_size), max(lds_block_size) "
                         "from counters_collection where kernel_name like '%render_rays%' group by kernel_name, counter_name").fetchall()
      for k, c, s, n, d, v, a, sc, lds in rows:
        print(f'  pmc {c:32s} sum={s:.6g} dispatches={n} per_dispatch={s/n:.6g}  (vgpr {v} agpr {a} scratch {sc} lds {lds})')
    except Exception as e:
      print('  (no counters)', e)
    try:
      rows = cur.execute("select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, scratch_size, lds_size from kernels where name like '%render_rays%'").fetchall()
      for r in rows[-3:]:
        print(f'  dispatch dur_ms={r[1]/1e6:.4f} grid={r[2]} wg={r[3]} vgpr={r[4]} agpr={r[5]} scratch={r[6]} lds={r[7]}')
    except Exception as e:
      print('  (no kernels view)', e)

if __name__ == '__main__':
  main(sys.argv[1])
