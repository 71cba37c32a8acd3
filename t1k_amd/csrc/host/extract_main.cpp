// fastq-extractor -- drop-in replacement of the reference's fastq-extractor executable for run-t1k ("$WD/fastq-extractor ...", run-t1k:377-403)
#include "../../../include/t1k_gpu.h"
int main(int argc, char **argv) { return t1k_extractor_main(argc, argv); }
