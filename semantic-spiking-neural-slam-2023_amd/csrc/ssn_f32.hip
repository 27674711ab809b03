// float instantiation of the step-loop kernels: the fast mode (FMA contraction allowed).
#include "ssn_kernels.hpp"
namespace ssn { SSN_INSTANTIATE(float) }
