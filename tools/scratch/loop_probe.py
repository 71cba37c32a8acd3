import os, sys, time, json
sys.path.insert(0, "/root/repo")
import bench, t1k_amd
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ref, pfx = bench.ensure_inputs("/tmp/t1k_bench", pairs, 24, 1.0, seed=2)
for pipes, batch in ((4, 16384), (2, 16384), (1, 16384), (4, 8192), (2, 8192), (4, 32768)):
    os.environ["T1K_PIPELINES"] = str(pipes)
    os.environ["T1K_BATCH"] = str(batch)
    job = t1k_amd.Job(ref, ref_seq_similarity=0.97)
    job.load_reads(pfx + "_1.fq", pfx + "_2.fq")
    for it in range(3):
        t0 = time.time(); job.run(); dt = time.time() - t0
        st = job.stats()
        print("pipes %d batch(frag) %d run %d: %.3f s; device loop %.0f ms coalesce %.0f em %.0f | kernel ms: seed %.0f chain %.0f ext %.0f sel %.0f full %.0f pair %.0f" % (
            pipes, batch, it, dt, st["ms_device"], st["ms_coalesce"], st["ms_em"], st["ms_seed"], st["ms_chain"], st["ms_extend"], st["ms_select"], st["ms_fullalign"], st["ms_pair"]), flush=True)
    job.close()
