// oracle/shim/melstreamer/atlcoll.h -- TEST INFRASTRUCTURE ONLY: CAtlMap lives in this directory's stdafx.h
#pragma once
#include "stdafx.h"
