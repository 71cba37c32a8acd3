// bam-extractor -- drop-in replacement of the reference's bam-extractor executable for run-t1k ("$WD/bam-extractor -b ...", run-t1k:350)
#include "../../../include/t1k_gpu.h"
int main(int argc, char **argv) { return t1k_bam_extractor_main(argc, argv); }
