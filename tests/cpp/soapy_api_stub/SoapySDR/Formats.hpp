// COMPILE-CHECK STUB -- see Device.hpp in this directory.
#pragma once
#define SOAPY_SDR_CF64 "CF64"
#define SOAPY_SDR_CF32 "CF32"
#define SOAPY_SDR_CS16 "CS16"
#define SOAPY_SDR_CS8 "CS8"
#define SOAPY_SDR_TX 0
#define SOAPY_SDR_RX 1
