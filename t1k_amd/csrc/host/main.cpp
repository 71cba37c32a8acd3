// genotyper -- drop-in replacement of the reference's genotyper executable for run-t1k ("$WD/genotyper ...", run-t1k:430,434)
#include "../../../include/t1k_gpu.h"
int main(int argc, char **argv) { return t1k_genotyper_main(argc, argv); }
