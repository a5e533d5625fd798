/* placeholder translation unit; hgt restatement lands here */
#include "orc_sampling.h"
