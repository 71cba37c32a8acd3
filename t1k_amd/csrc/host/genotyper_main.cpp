// t1k_amd/csrc/host/genotyper_main.cpp -- t1k_genotyper_main(): the argv-compatible replacement of the reference's genotyper executable
// (Genotyper.cpp:194-738, invoked by run-t1k:430,434) on top of the job layer.
#include "job_internal.h"

extern "C" {

// ------------------------------------------------------------------------------------------------------------------
// the executable's entry point
// ------------------------------------------------------------------------------------------------------------------
}  // extern "C"
namespace t1k {
void logLine(const char *fmt, ...) {  // same shape as the reference's PrintLog (Genotyper.cpp:113-124): users grep these lines
  char msg[4096];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  time_t now = time(NULL);
  char stamp[128];
  strftime(stamp, sizeof(stamp), "%c", localtime(&now));
  fprintf(stderr, "[%s] %s\n", stamp, msg);
}
}  // namespace t1k
extern "C" {

static const char *kUsage =
    "./genotyper [OPTIONS]:   (MI355X build of the T1K genotyper stage; same options as the reference)\n"
    "Required:\n"
    "\t-f STRING: fasta file with the allele reference sequences\n"
    "\t-u STRING: single-end read file, or\n"
    "\t-1 STRING -2 STRING: paired-end read files\n"
    "Optional:\n"
    "\t-a STRING: abundance file (skips the EM)\n"
    "\t-t INT: host threads (default: 1)\n"
    "\t-o STRING: output prefix (default: t1k)\n"
    "\t-n INT: maximal number of alleles per read (default: 2000)\n"
    "\t-s FLOAT: minimum alignment similarity (default: 0.8)\n"
    "\t--alleleWhitelist STRING: only consider reads aligned to the listed allele series\n"
    "\t--barcode STRING: barcode file\n"
    "\t--frac FLOAT: filter alleles below this fraction of the dominant allele (default: 0.15)\n"
    "\t--cov FLOAT: filter genes with average coverage below this value (default: 1.0)\n"
    "\t--crossGeneRate FLOAT: contribution of other genes' expression (default: 0.04)\n"
    "\t--relaxIntronAlign: allow one more mismatch in intronic alignment\n"
    "\t--alleleDigitUnits INT: number of name units in the genotyping result (default: automatic)\n"
    "\t--alleleDelimiter CHR: delimiter of the name units (default: automatic)\n"
    "\t--outputReadAssignment: write prefix_assign.tsv\n"
    "\t--squaremMinAlpha FLOAT: lower bound (negative) of the SQUAREM step length\n"
    "\t--device INT: GPU ordinal (default: $T1K_DEVICE or 0)\n"
    "\t--gpus INT: shard the fragments over the first INT GPUs ($T1K_GPUS=0,1,.. names them; a GPU may be named twice)\n";

int t1k_genotyper_main(int argc, char **argv) {
  if (argc <= 1) { fprintf(stderr, "%s", kUsage); return 0; }  // Genotyper.cpp:199-203
  const double tMain = nowMs();
  static struct option longOpts[] = {{"frac", required_argument, 0, 1000}, {"cov", required_argument, 0, 1001}, {"crossGeneRate", required_argument, 0, 1002},
                                     {"barcode", required_argument, 0, 1003}, {"relaxIntronAlign", no_argument, 0, 1004},
                                     {"alleleDigitUnits", required_argument, 0, 1005}, {"alleleDelimiter", required_argument, 0, 1006},
                                     {"alleleWhitelist", required_argument, 0, 1007}, {"outputReadAssignment", no_argument, 0, 1008},
                                     {"squaremMinAlpha", required_argument, 0, 1009}, {"device", required_argument, 0, 1010}, {"gpus", required_argument, 0, 1011}, {0, 0, 0, 0}};
  t1k_job_params p;
  t1k_job_params_default(&p);
  if (const char *d = getenv("T1K_DEVICE")) p.device = atoi(d);
  int nGpus = 0;
  std::string refFile, prefix = "t1k", barcode, whitelistFile, abundance;
  std::vector<const char *> f1, f2, single;  // every -u / -1 / -2 counts: the files are read back to back (ReadFiles::AddReadFile)
  optind = 1;
  int c, idx = 0;
  while ((c = getopt_long(argc, argv, "f:a:u:1:2:o:t:n:s:b:", longOpts, &idx)) != -1) {
    switch (c) {
      case 'f': refFile = optarg; break;
      case 'a': abundance = optarg; break;
      case 'u': single.push_back(optarg); break;
      case '1': f1.push_back(optarg); break;
      case '2': f2.push_back(optarg); break;
      case 'o': prefix = optarg; break;
      case 't': p.threads = atoi(optarg); break;
      case 'n': p.dev.max_assign_cnt = atoi(optarg); break;
      case 's': p.dev.ref_seq_similarity = atof(optarg); break;
      case 'b': break;
      case 1000: p.filter_frac = atof(optarg); break;
      case 1001: p.filter_cov = atof(optarg); break;
      case 1002: p.cross_gene_rate = atof(optarg); break;
      case 1003: barcode = optarg; break;
      case 1004: p.dev.relax_intron_align = 1; break;
      case 1005: p.allele_digit_units = atoi(optarg); break;
      case 1006: p.allele_delimiter = optarg[0]; break;
      case 1007: whitelistFile = optarg; break;
      case 1008: p.output_read_assignment = 1; break;
      case 1009: p.squarem_min_alpha = atof(optarg); break;
      case 1010: p.device = atoi(optarg); break;
      case 1011: nGpus = atoi(optarg); break;
      default: fprintf(stderr, "%s", kUsage); return EXIT_FAILURE;
    }
  }
  if (refFile.empty()) { fprintf(stderr, "Need to use -f to specify the reference sequences.\n"); return EXIT_FAILURE; }
  if (p.dev.max_assign_cnt == 0) p.dev.max_assign_cnt = -1;  // "-n 0" disables the cap in the reference (maxAssignCnt > 0 test)
  // GPUs of the job: --gpus N = the first N devices, T1K_GPUS = an explicit list; one rank (thread, job, context set) per entry
  std::vector<int> devices;
  if (const char *e = getenv("T1K_GPUS")) {
    for (const char *q = e; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
  } else if (nGpus > 1) {
    for (int d = 0; d < nGpus; ++d) devices.push_back(d);
  }
  if (devices.empty()) devices.push_back(p.device);
  const int R = (int)devices.size();
  std::vector<t1k_job *> jobs(R, nullptr);
  std::vector<int> rcs(R, T1K_OK);
  auto destroyAll = [&] { for (t1k_job *j : jobs) t1k_job_destroy(j); };
  const bool paired = !f2.empty();
  const std::vector<const char *> &first = !f1.empty() ? f1 : single;
  // T1K_SHARD_INPUT=1: every rank indexes only its own fragments and writes only its own part of the *_aligned*.fa files, as ranks
  // in separate processes do (bench.py under torchrun); by default the ranks of this process share one index built by all host threads
  const bool shardInput = R > 1 && getenv("T1K_SHARD_INPUT") && atoi(getenv("T1K_SHARD_INPUT")) != 0;
  // the read files are mapped and indexed while the reference is parsed and the contexts come up (the reference's main does the two
  // one after the other, Genotyper.cpp:226-232 and 365-454; neither needs the other)
  t1k_reads *opened = nullptr;
  int rcOpen = T1K_OK;
  std::thread opener;
  if (!shardInput && !first.empty() && !getenv("T1K_SERIAL_OPEN"))
    opener = std::thread([&] {
      // (one rank: an ordinary .gz input -- the barcode file with it -- is handed to the loop while it is still being inflated)
      if (R == 1) rcOpen = t1k_reads_open_stream(first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str(), p.threads, &opened);
      else rcOpen = t1k_reads_open(first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str(), p.threads, &opened);
    });
  {
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r)
      th.emplace_back([&, r] { t1k_job_params q = p; q.device = devices[r]; rcs[r] = t1k_job_create(&q, refFile.c_str(), &jobs[r]); });
    for (auto &t : th) t.join();
  }
  const bool openedBeside = opener.joinable();
  if (openedBeside) opener.join();
  for (int r = 0; r < R; ++r)
    if (rcs[r] != T1K_OK) {
      fprintf(stderr, "genotyper: %s\n", jobs[r] ? t1k_job_last_error(jobs[r]) : "initialisation failed");
      if (jobs[r] && jobs[r]->ref.al.empty()) fprintf(stderr, "Need to use -f to specify the reference sequences.\n");
      destroyAll();
      t1k_reads_close(opened);
      return EXIT_FAILURE;
    }
  t1k_job *job = jobs[0];
  if (!whitelistFile.empty()) {  // Genotyper::SetAlleleWhitelist (Genotyper.hpp:684-705): whole major-allele series
    FILE *fp = fopen(whitelistFile.c_str(), "r");
    if (!fp) { fprintf(stderr, "genotyper: cannot open %s\n", whitelistFile.c_str()); destroyAll(); t1k_reads_close(opened); return EXIT_FAILURE; }
    std::set<int> majors;
    std::map<std::string, int> majorId;
    for (size_t i = 0; i < job->ref.majorName.size(); ++i) majorId[job->ref.majorName[i]] = (int)i;
    char name[512];
    while (fscanf(fp, "%511s", name) == 1) {
      std::string g, m;
      job->ref.splitName(name, g, m, 0);
      auto it = majorId.find(m);
      if (it != majorId.end()) majors.insert(it->second);
    }
    fclose(fp);
    for (t1k_job *j : jobs) {
      j->whitelist.assign(j->ref.al.size(), 0);
      for (size_t a = 0; a < j->ref.al.size(); ++a) j->whitelist[a] = majors.count(j->ref.al[a].major) ? 1 : 0;
    }
  }
  for (t1k_job *j : jobs) j->abundanceFile = abundance;
  if (first.empty()) { fprintf(stderr, "genotyper: no read file given (-u, or -1 and -2)\n"); destroyAll(); t1k_reads_close(opened); return EXIT_FAILURE; }
  auto loadInto = [&](t1k_job *j) {
    return t1k_job_load_reads_multi(j, first.data(), (uint32_t)first.size(), paired ? f2.data() : nullptr, (uint32_t)f2.size(), barcode.empty() ? nullptr : barcode.c_str());
  };
  int rc = T1K_OK;
  bool foundLater = false;
  if (!shardInput) {
    if (openedBeside) { rc = t1k_job_attach_reads(job, opened); opened = nullptr; if (rc == T1K_OK) rc = rcOpen; }  // (a failed open: the handle carries the message into the job)
    else rc = loadInto(job);
    if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); destroyAll(); return EXIT_FAILURE; }
    job->in->dropInflatedText = R == 1 && !getenv("T1K_KEEP_TEXT");  // this process runs the job once: the text of written fragments is not needed again
    foundLater = job->in->streaming;  // (a streamed input: counted when the stream has ended, i.e. behind the loop)
    if (!foundLater) logLine("Found %d read fragments. Start read assignment.", (int)job->in->nAll());
    t1k_job_set_output_prefix(job, prefix.c_str());  // the aligned-read files are written while the EM runs
  }
  if (R == 1) rc = t1k_job_run(job);
  else {
    // one thread per rank: the ranks meet in the collectives of t1k_job_run (RCCL when every rank has its own GPU)
    t1k_comm_group *group = t1k_comm_group_create(R);
    std::vector<t1k_comm *> comms(R, nullptr);
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r)
      th.emplace_back([&, r] {
        int x = (r && !shardInput) ? t1k_job_share_reads(jobs[r], job) : T1K_OK;
        const int y = t1k_comm_init(t1k_job_ctx(jobs[r]), R, r, nullptr, group, -1, &comms[r]);  // collective: every rank calls it
        if (x == T1K_OK && y != T1K_OK) { jobs[r]->err = comms[r] ? t1k_comm_last_error(comms[r]) : "cannot create the communicator"; x = y; }
        if (x == T1K_OK) x = t1k_job_set_shard(jobs[r], r, R, comms[r]);
        if (x == T1K_OK && shardInput) {
          x = loadInto(jobs[r]);  // collective
          if (x == T1K_OK) {
            if (r == 0) logLine("Found %d read fragments. Start read assignment.", (int)jobs[r]->in->nAll());
            t1k_job_set_output_prefix(jobs[r], prefix.c_str());
          }
        }
        rcs[r] = x == T1K_OK ? t1k_job_run(jobs[r]) : x;
        if (rcs[r] == T1K_OK && shardInput && r) rcs[r] = t1k_job_write_outputs(jobs[r], prefix.c_str());  // its part of the read files (rank 0: below)
        // a rank that gives up must not leave the others waiting at the next exchange: they are released with an error of their own
        if (rcs[r] != T1K_OK && comms[r]) (void)t1k_comm_abort(comms[r]);
      });
    for (auto &t : th) t.join();
    for (int pass = 0; pass < 2 && rc == T1K_OK; ++pass)  // report the rank that failed, not the ones it released (T1K_ERR_STATE)
      for (int r = 0; r < R && rc == T1K_OK; ++r)
        if (rcs[r] != T1K_OK && (pass == 1 || rcs[r] != T1K_ERR_STATE)) { rc = rcs[r]; if (r) job->err = t1k_job_last_error(jobs[r]); }
    for (t1k_comm *c : comms) t1k_comm_destroy(c);
    t1k_comm_group_destroy(group);
  }
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); destroyAll(); return EXIT_FAILURE; }
  if (foundLater) logLine("Found %d read fragments. Start read assignment.", (int)job->in->nAll());
  logLine("Finish read end assignments.");
  const double groups = (double)job->gt.nGroups();
  logLine("Finish read fragment assignments. %d read fragments can be assigned (average %.2lf alleles/read).", (int)job->gt.assignedFragments,
          job->gt.sumAssign / groups);
  if (abundance.empty()) logLine("Finish allele quantification in %d EM iterations.", job->gt.emIterations);
  rc = t1k_job_write_outputs(job, prefix.c_str());
  if (rc != T1K_OK) { fprintf(stderr, "genotyper: %s\n", t1k_job_last_error(job)); destroyAll(); return EXIT_FAILURE; }
  logLine("Genotyping finishes.");
  const double tOut = nowMs();
  destroyAll();
  if (getenv("T1K_DEBUG_PHASES")) {  // (what a stopwatch around the process sees beyond this: loading the executable and the HIP runtime before main, the exit behind it)
    fprintf(stderr, "[t1k job] main: %.1f ms from its first line to the outputs, %.1f ms to release the job\n", tOut - tMain, nowMs() - tOut);
    // what the process still maps when it leaves (the kernel takes the address space apart before the parent sees the exit)
    if (FILE *fp = fopen("/proc/self/smaps_rollup", "r")) {
      char line[256];
      std::string all;
      while (fgets(line, sizeof line, fp))
        if (!strncmp(line, "Rss:", 4) || !strncmp(line, "Anonymous:", 10) || !strncmp(line, "Shared_Clean:", 13) || !strncmp(line, "Shared_Dirty:", 13) || !strncmp(line, "Private_Clean:", 14) ||
            !strncmp(line, "Private_Dirty:", 14) || !strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Locked:", 7)) {
          std::string l(line);
          while (!l.empty() && (l.back() == '\n' || l.back() == ' ')) l.pop_back();
          size_t a = l.find(':');
          size_t b = l.find_first_not_of(' ', a + 1);
          all += l.substr(0, a + 1) + " " + (b == std::string::npos ? "" : l.substr(b)) + "; ";
        }
      fclose(fp);
      fprintf(stderr, "[t1k job] address space at the end of main: %s\n", all.c_str());
    }
    if (getenv("T1K_DEBUG_MAPS"))  // the largest resident mappings (what the exit has to take apart page by page)
      if (FILE *fp = fopen("/proc/self/smaps", "r")) {
        struct Reg { std::string head; unsigned long rss = 0, anon = 0; };
        std::vector<Reg> regs;
        char line[512];
        while (fgets(line, sizeof line, fp)) {
          unsigned long a, b;
          if (sscanf(line, "%lx-%lx ", &a, &b) == 2 && strchr(line, '-') && (strstr(line, " r") || strstr(line, " -"))) { Reg r; r.head = line; while (!r.head.empty() && r.head.back() == '\n') r.head.pop_back(); regs.push_back(r); }
          else if (!regs.empty() && !strncmp(line, "Rss:", 4)) regs.back().rss = strtoul(line + 4, nullptr, 10);
          else if (!regs.empty() && !strncmp(line, "Anonymous:", 10)) regs.back().anon = strtoul(line + 10, nullptr, 10);
        }
        fclose(fp);
        std::sort(regs.begin(), regs.end(), [](const Reg &x, const Reg &y) { return x.rss > y.rss; });
        for (size_t i = 0; i < regs.size() && i < 24; ++i) fprintf(stderr, "[t1k job]   rss %8lu kB (anonymous %8lu kB)  %s\n", regs[i].rss, regs[i].anon, regs[i].head.c_str());
      }
    if (FILE *fp = fopen("/proc/self/status", "r")) {
      char line[256];
      while (fgets(line, sizeof line, fp))
        if (!strncmp(line, "Threads:", 8) || !strncmp(line, "VmPeak:", 7) || !strncmp(line, "VmHWM:", 6) || !strncmp(line, "VmPTE:", 6)) fprintf(stderr, "[t1k job]   %s", line);
      fclose(fp);
    }
  }
  return 0;
}

}  // extern "C"
