// <fss/prg/aes128_mmo.cuh>: the include line a caller of myl7/fss writes for fss::prg::Aes128Mmo<mul>.
// Kept so that such a caller's #include lines compile unchanged against this library ("include lines");
// nothing is defined here -- the declarations live in fss_amd.hpp.
#pragma once
#include "../../fss_amd.hpp"
