// Second translation unit of the host module: negative sampling and HGT sampling bindings.
#include <torch/extension.h>

#include "../../include/tchgeo.h"

namespace py = pybind11;

void register_more(py::module_ &m) { (void)m; }
