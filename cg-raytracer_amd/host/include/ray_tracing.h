// Forwarder: the surface of the reference header of this name lives in cgrt_host_api.h.
#pragma once
#include "cgrt_host_api.h"
