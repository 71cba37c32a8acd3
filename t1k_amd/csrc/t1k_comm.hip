// t1k_amd/csrc/t1k_comm.hip -- multi-GPU exchange steps of the genotyper stage over RCCL (filled in below)
#include "t1k_dev.h"
#include "t1k_launch.h"
