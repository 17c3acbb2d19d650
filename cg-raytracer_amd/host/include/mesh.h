// Forwarder: the types of the reference header of this name live in cgrt_host_types.h.
#pragma once
#include "cgrt_host_types.h"
