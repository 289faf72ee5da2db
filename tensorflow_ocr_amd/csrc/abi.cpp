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
    case OCR_ERR_RCCL: return "RCCL error or RCCL not available (ocr_comm_last_error)";
    default: return "unknown status";
  }
}


// CRC-32C (Castagnoli, reflected 0x82F63B78), slicing-by-8, HOST routine: the checksum TensorFlow
// checkpoint bundles carry per tensor and per table block (tensorflow_ocr_amd/tf_bundle.py).
// Semantics of LevelDB's crc32c::Extend(seed, data, n).
extern "C" uint32_t ocr_crc32c(const void* data, size_t n, uint32_t seed) {
  static uint32_t tab[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
    ready = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = seed ^ 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t lo, hi;
    __builtin_memcpy(&lo, p, 4);
    __builtin_memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
        tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
