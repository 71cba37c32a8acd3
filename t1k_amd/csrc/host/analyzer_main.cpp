// t1k_amd/csrc/host/analyzer_main.cpp -- the `analyzer` executable: argv -> t1k_analyzer_main (libt1k_gpu.so, host/job.cpp)
#include "../../../include/t1k_gpu.h"
int main(int argc, char **argv) { return t1k_analyzer_main(argc, argv); }
