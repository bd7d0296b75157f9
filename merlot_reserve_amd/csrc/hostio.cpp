// Host-side record I/O of the real-data input path (include/mreserve_hip.h: mr_crc32c, mr_tfrecord_scan): the container format of the
// reference's training shards (pretrain/dataloader.py:884 tf.data.TFRecordDataset).  Plain C++, no GPU: the reference reads these files through
// TensorFlow's C++ record reader, this is its counterpart for merlot_reserve_amd/records.py.
//
// A TFRecord file is a sequence of   uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)
// (little endian), masked_crc = rotr(crc, 15) + 0xa282ead8; CRC-32C is the Castagnoli polynomial 0x1EDC6F41 (reflected 0x82F63B78).
#include <stdint.h>
#include <string.h>
#include "../../include/mreserve_hip.h"

void mr_set_error(const char* fmt, ...);

namespace {

struct Crc32cTable {
    uint32_t t[8][256];
    Crc32cTable() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
    }
};
const Crc32cTable g_tab;

// slicing-by-8: eight table lookups per 8 input bytes (~1.5-2 GB/s on one core; a base shard of ~1.5 MB records checks in about a millisecond each)
uint32_t crc32c_update(uint32_t crc, const uint8_t* p, size_t n) {
    crc = ~crc;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) { crc = g_tab.t[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        const uint32_t lo = (uint32_t)w ^ crc, hi = (uint32_t)(w >> 32);
        crc = g_tab.t[7][lo & 0xffu] ^ g_tab.t[6][(lo >> 8) & 0xffu] ^ g_tab.t[5][(lo >> 16) & 0xffu] ^ g_tab.t[4][lo >> 24] ^
              g_tab.t[3][hi & 0xffu] ^ g_tab.t[2][(hi >> 8) & 0xffu] ^ g_tab.t[1][(hi >> 16) & 0xffu] ^ g_tab.t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = g_tab.t[0][(crc ^ *p++) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

}  // namespace

extern "C" uint32_t mr_crc32c(const void* data, int64_t n, uint32_t crc) {
    if (data == nullptr || n <= 0) return crc;
    return crc32c_update(crc, static_cast<const uint8_t*>(data), (size_t)n);
}

extern "C" uint32_t mr_crc32c_masked(const void* data, int64_t n) { return mask_crc(mr_crc32c(data, n, 0u)); }

extern "C" int64_t mr_tfrecord_scan(const void* buf, int64_t n, int64_t* offsets, int64_t* lengths, int64_t cap, int32_t verify) {
    if (buf == nullptr || n < 0 || cap < 0 || (cap > 0 && (offsets == nullptr || lengths == nullptr))) {
        mr_set_error("mr_tfrecord_scan: bad arguments");
        return MR_EINVAL;
    }
    const uint8_t* p = static_cast<const uint8_t*>(buf);
    int64_t pos = 0, count = 0;
    while (pos < n) {
        if (n - pos < 12) { mr_set_error("mr_tfrecord_scan: truncated header at byte %ld of %ld", (long)pos, (long)n); return MR_EINVAL; }
        uint64_t len;
        uint32_t lcrc;
        memcpy(&len, p + pos, 8);
        memcpy(&lcrc, p + pos + 8, 4);
        if (verify && mask_crc(crc32c_update(0u, p + pos, 8)) != lcrc) {
            mr_set_error("mr_tfrecord_scan: record %ld: corrupted length field at byte %ld", (long)count, (long)pos);
            return MR_EINVAL;
        }
        // signed first: a shard cut 12-15 bytes into a record leaves n - pos - 16 negative, which the cast below would turn into a huge bound
        if (n - pos < 16 || len > (uint64_t)(n - pos - 16)) {
            mr_set_error("mr_tfrecord_scan: record %ld at byte %ld: %llu data bytes run past the end of the buffer", (long)count, (long)pos, (unsigned long long)len);
            return MR_EINVAL;
        }
        const uint8_t* d = p + pos + 12;
        if (verify) {
            uint32_t dcrc;
            memcpy(&dcrc, d + len, 4);
            if (mask_crc(crc32c_update(0u, d, (size_t)len)) != dcrc) {
                mr_set_error("mr_tfrecord_scan: record %ld at byte %ld: data checksum mismatch", (long)count, (long)pos);
                return MR_EINVAL;
            }
        }
        if (count < cap) { offsets[count] = pos + 12; lengths[count] = (int64_t)len; }
        ++count;
        pos += 16 + (int64_t)len;
    }
    return count;
}
