# times every API call of one bench step (second step = warm pool), with the library's own phase lines
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["T1K_DEBUG_PHASES"] = "1"
import bench, t1k_amd
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
ref, pfx = bench.ensure_inputs("/tmp/t1k_bench", pairs, 24, 1.0, seed=2)
for step in range(2):
    t = [time.perf_counter()]
    job = t1k_amd.Job(ref, ref_seq_similarity=0.97, device=0); t.append(time.perf_counter())
    job.load_reads([pfx + "_1.fq"], [pfx + "_2.fq"]); t.append(time.perf_counter())
    job.set_output_prefix("/tmp/t1k_bench/probe"); job.run(); t.append(time.perf_counter())
    job.write_outputs("/tmp/t1k_bench/probe"); t.append(time.perf_counter())
    st = job.stats(); job.genotype_text(); t.append(time.perf_counter())
    job.close(); t.append(time.perf_counter())
    names = ["create", "load_reads", "run", "write_outputs", "stats+text", "close"]
    sys.stderr.write("[probe] step %d: total %.0f ms | " % (step, (t[-1] - t[0]) * 1e3) + ", ".join("%s %.0f" % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])) + "\n")
