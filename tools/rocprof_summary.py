"""Dump the kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite) as text.

    python tools/rocprof_summary.py gpurun_out/prof_a profiles/r01_kernel_stats.txt "command line"
"""
import glob
import sqlite3
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    dbs = glob.glob(d + '/**/*.db', recursive=True)
    lines = ['# rocprofv3 --kernel-trace --stats summary (durations in us)', '# ' + note]
    for db in dbs:
        con = sqlite3.connect(db)
        cur = con.cursor()
        lines.append('# source: ' + db)
        lines.append('%-90s %8s %16s %16s %8s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for name, calls, total, avg, pct in cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            lines.append('%-90s %8d %16.0f %16.0f %8.3f' % (name[:90], calls, total, avg, pct))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:12]))


if __name__ == '__main__':
    main()
