"""Per-kernel sums of the counters collected by tools/sq_counters.sh (rocpd databases)."""
import collections
import glob
import sqlite3
import sys


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for d in sys.argv[1:]:
        for path in glob.glob(d + "/**/*.db", recursive=True):
            c = sqlite3.connect(path)
            seen = set()
            for disp, name, cn, v in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
                k = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("sebm25::", "")
                per[k][cn] += v
                if (disp, cn) not in seen and cn in ("SQ_WAVES", "SQ_INSTS_SALU"):
                    launches[(k, cn)] += 1
                seen.add((disp, cn))
    names = sorted({cn for k in per for cn in per[k]})
    for k in sorted(per):
        if not k.startswith(("k_sort_hist", "k_sort_scatter", "k_postings<", "k_score", "k_merge")):
            continue
        n = max(launches[(k, "SQ_WAVES")], launches[(k, "SQ_INSTS_SALU")], 1)
        print(k, "launches", n)
        for cn in names:
            if cn in per[k]:
                print("   %-24s %.4g per launch" % (cn, per[k][cn] / n))


if __name__ == "__main__":
    main()
