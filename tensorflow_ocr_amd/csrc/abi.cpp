// ABI bookkeeping for libocr_hip.so (include/ocr_hip.h).
#include "common.h"

extern "C" int ocr_abi_version(void) { return OCR_ABI_VERSION; }

extern "C" const char* ocr_storage_dtype(void) { return OCR_STORAGE_NAME; }

extern "C" const char* ocr_status_string(int status) {
  switch (status) {
    case OCR_OK: return "ok";
    case OCR_ERR_INVALID_ARG: return "invalid argument";
    case OCR_ERR_UNSUPPORTED: return "unsupported shape for this kernel";
    case OCR_ERR_HIP: return "HIP runtime error";
    case OCR_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

