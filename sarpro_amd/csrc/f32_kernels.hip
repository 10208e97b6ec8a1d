// f32_kernels.hip -- kernels of the f32-input flavour (pending)
#include "kernels.h"
