// Third translation unit of the host module: hgt_sampling (python.rs:399-482).
#include "host_common.h"

void register_hgt(py::module_ &m) { (void)m; }
