// ORACLE (test infrastructure only -- never linked into the product path).
// C entry points of liboracle.so, bound with ctypes from tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg ONLY.  See oracle/README.md.
#pragma once
#include <cstdint>
extern "C" {
void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int thFAST);
void orc_orb_destroy(void* h);
void orc_orb_tables(void* h, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2, int* quota, int* umax16);
int orc_orb_extract(void* h, const uint8_t* img, int w, int hh, int stride, void* kps_out, uint8_t* desc_out, int cap);
}
