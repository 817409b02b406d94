// Host side of the batch-of-files driver: a pool of I/O threads that moves the int16 frames of 16-bit PCM wav
// files between the file system and PINNED staging memory -- what `scipy.io.wavfile.read` / `.write` do in every script of
// the reference (examples/dsd100/separate_dsd.py:275-282 and :307-309), minus the float detour: the division by 32767, the
// mix-down and the int16 conversion run on the device (dcs_pcm16_to_float / dcs_pcm_to_int16).
//
// No device code in this file.  It exists because the Python driver's thread pool spent more time handing the interpreter
// lock around (0.1 ms per submitted task with 16 workers, examples/separate_batch.py --stats) than the device spends on a
// file; here a batch of reads or writes is ONE call that returns at once, the threads never touch the interpreter, and the
// caller collects the batch when it needs the frames (reads) or the staging block back (writes).
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dcs_internal.h"

#define DCS_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) DCS_FAIL(code, __VA_ARGS__); \
    } while (0)

struct dcs_wav_batch {
    std::mutex m;
    std::condition_variable cv;
    int remaining = 0;
    std::vector<std::string> paths;
};

namespace {

struct WavTask {
    dcs_wav_batch* batch;
    int index;
    bool write;
    // read
    void* dst;
    int64_t cap;
    int32_t* rate_out;
    int64_t* frames_out;
    int32_t* channels_out;
    // write
    const int16_t* data;
    int64_t n_frames;
    int32_t channels, rate;
    int32_t* status;
};

uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
void put32(unsigned char* p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
void put16(unsigned char* p, uint16_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

bool pread_all(int fd, void* buf, size_t n, off_t off) {
    char* p = static_cast<char*>(buf);
    while (n) {
        const ssize_t k = pread(fd, p, n, off);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        p += k;
        off += k;
        n -= (size_t)k;
    }
    return true;
}

// 0: frames read | 1: not plain 16-bit PCM, truncated header, or larger than the staging slot (the caller takes the scripts'
// float path for this file) | -errno: the file could not be opened or read
int read_pcm16(const char* path, void* dst, int64_t cap, int32_t* rate_out, int64_t* frames_out, int32_t* channels_out) {
    static const unsigned char kPcmSub[16] = {0x01, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71};
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -errno;
    struct stat st;
    if (fstat(fd, &st) != 0) { const int e = errno; close(fd); return -e; }
    const int64_t fsize = st.st_size;
    unsigned char h[64];
    int rc = 1;
    int64_t pos = 12;
    int64_t rate = -1, channels = 0;
    if (fsize >= 12 && pread_all(fd, h, 12, 0) && memcmp(h, "RIFF", 4) == 0 && memcmp(h + 8, "WAVE", 4) == 0) {
        for (;;) {                                            // chunk walk, as scipy's reader does it
            if (pos + 8 > fsize || !pread_all(fd, h, 8, pos)) break;
            const int64_t size = le32(h + 4);
            pos += 8;
            if (memcmp(h, "fmt ", 4) == 0) {
                if (size < 16) break;
                const size_t take = (size_t)(size < 40 ? size : 40);
                if (pos + (int64_t)take > fsize || !pread_all(fd, h, take, pos)) break;
                int code = le16(h);
                channels = le16(h + 2);
                rate = le32(h + 4);
                const int align = le16(h + 12), bits = le16(h + 14);
                if (code == 0xFFFE && size >= 40 && memcmp(h + 24, kPcmSub, 16) == 0) code = 1;
                if (code != 1 || bits != 16 || channels < 1 || align != 2 * channels) break;
                pos += size + (size & 1);
            } else if (memcmp(h, "data", 4) == 0) {
                if (rate < 0) break;
                int64_t nbytes = size < fsize - pos ? size : fsize - pos;
                const int64_t frames = nbytes / (2 * channels);
                nbytes = frames * 2 * channels;
                if (nbytes > cap) break;
                if (nbytes > 0 && !pread_all(fd, dst, (size_t)nbytes, pos)) { rc = -EIO; break; }
                *rate_out = (int32_t)rate;
                *frames_out = frames;
                *channels_out = (int32_t)channels;
                rc = 0;
                break;
            } else {
                pos += size + (size & 1);
            }
        }
    }
    close(fd);
    return rc;
}

void make_parents(const std::string& path) {
    for (size_t i = 1; i < path.size(); ++i)
        if (path[i] == '/') {
            const std::string dir = path.substr(0, i);
            mkdir(dir.c_str(), 0777);                       // EEXIST is the common case
        }
}

// the 44 bytes scipy.io.wavfile.write puts in front of int16 data, then the frames: one writev
int write_pcm16(const std::string& path, const int16_t* data, int64_t n_frames, int channels, int rate) {
    int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0 && errno == ENOENT) {
        make_parents(path);
        fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    }
    if (fd < 0) return -errno;
    const uint64_t nbytes = (uint64_t)n_frames * 2 * (uint64_t)channels;
    unsigned char hd[44];
    memcpy(hd, "RIFF", 4);
    put32(hd + 4, (uint32_t)(36 + nbytes));
    memcpy(hd + 8, "WAVEfmt ", 8);
    put32(hd + 16, 16);
    put16(hd + 20, 1);
    put16(hd + 22, (uint16_t)channels);
    put32(hd + 24, (uint32_t)rate);
    put32(hd + 28, (uint32_t)rate * 2u * (uint32_t)channels);
    put16(hd + 32, (uint16_t)(2 * channels));
    put16(hd + 34, 16);
    memcpy(hd + 36, "data", 4);
    put32(hd + 40, (uint32_t)nbytes);
    struct iovec iov[2] = {{hd, 44}, {const_cast<int16_t*>(data), (size_t)nbytes}};
    size_t done = 0;
    const size_t total = 44 + (size_t)nbytes;
    int rc = 0;
    while (done < total) {
        struct iovec cur[2];
        int n = 0;
        size_t skip = done;
        for (int i = 0; i < 2; ++i) {
            if (skip >= iov[i].iov_len) { skip -= iov[i].iov_len; continue; }
            cur[n].iov_base = static_cast<char*>(iov[i].iov_base) + skip;
            cur[n].iov_len = iov[i].iov_len - skip;
            skip = 0;
            ++n;
        }
        const ssize_t k = writev(fd, cur, n);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) { rc = k < 0 ? -errno : -EIO; break; }
        done += (size_t)k;
    }
    if (close(fd) != 0 && rc == 0) rc = -errno;
    return rc;
}

}  // namespace

struct dcs_wav_pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv;
    std::deque<WavTask> queue;
    bool stop = false;

    void run() {
        for (;;) {
            WavTask t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;                  // stop, and nothing left to do
                t = queue.front();
                queue.pop_front();
            }
            const std::string& path = t.batch->paths[t.index];
            const int rc = t.write ? write_pcm16(path, t.data, t.n_frames, t.channels, t.rate)
                                   : read_pcm16(path.c_str(), t.dst, t.cap, t.rate_out, t.frames_out, t.channels_out);
            if (t.status) *t.status = rc;
            {
                std::lock_guard<std::mutex> lk(t.batch->m);
                if (--t.batch->remaining == 0) t.batch->cv.notify_all();
            }
        }
    }
    void push(std::vector<WavTask>& tasks) {
        {
            std::lock_guard<std::mutex> lk(m);
            for (auto& t : tasks) queue.push_back(t);
        }
        cv.notify_all();
    }
};

extern "C" {

DCS_API int dcs_wav_pool_create(int n_threads, dcs_wav_pool** out) {
    DCS_REQUIRE(out != nullptr && n_threads >= 1 && n_threads <= 1024, DCS_EINVAL, "dcs_wav_pool_create: 1 .. 1024 threads");
    dcs_wav_pool* p = new dcs_wav_pool;
    try {
        for (int i = 0; i < n_threads; ++i) p->threads.emplace_back([p] { p->run(); });
    } catch (...) {
        {
            std::lock_guard<std::mutex> lk(p->m);
            p->stop = true;
        }
        p->cv.notify_all();
        for (auto& t : p->threads) t.join();
        delete p;
        dcs_set_error("dcs_wav_pool_create: could not start %d threads", n_threads);
        return DCS_ENOMEM;
    }
    *out = p;
    return DCS_OK;
}

DCS_API void dcs_wav_pool_destroy(dcs_wav_pool* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->stop = true;
    }
    p->cv.notify_all();
    for (auto& t : p->threads) t.join();                    // the queue is drained first: enqueued batches complete
    delete p;
}

DCS_API int dcs_wav_read_pcm16_async(dcs_wav_pool* p, int n, const char* const* paths, void* const* dst_h, const int64_t* cap,
                                     int32_t* rate, int64_t* n_frames, int32_t* channels, int32_t* status, dcs_wav_batch** out) {
    DCS_REQUIRE(p && out && n >= 0 && (n == 0 || (paths && dst_h && cap && rate && n_frames && channels && status)), DCS_EINVAL,
                "dcs_wav_read_pcm16_async: null argument");
    dcs_wav_batch* b = new dcs_wav_batch;
    b->remaining = n;
    std::vector<WavTask> tasks((size_t)n);
    for (int i = 0; i < n; ++i) {
        b->paths.emplace_back(paths[i] ? paths[i] : "");
        status[i] = 1;
        WavTask& t = tasks[(size_t)i];
        t = WavTask{};
        t.batch = b;
        t.index = i;
        t.write = false;
        t.dst = dst_h[i];
        t.cap = cap[i];
        t.rate_out = rate + i;
        t.frames_out = n_frames + i;
        t.channels_out = channels + i;
        t.status = status + i;
    }
    p->push(tasks);
    *out = b;
    return DCS_OK;
}

DCS_API int dcs_wav_write_pcm16_async(dcs_wav_pool* p, int n, const char* const* paths, const int16_t* const* data_h,
                                      const int64_t* n_frames, const int32_t* channels, const int32_t* rate, int32_t* status,
                                      dcs_wav_batch** out) {
    DCS_REQUIRE(p && out && n >= 0 && (n == 0 || (paths && data_h && n_frames && channels && rate && status)), DCS_EINVAL,
                "dcs_wav_write_pcm16_async: null argument");
    for (int i = 0; i < n; ++i)
        DCS_REQUIRE(paths[i] && n_frames[i] >= 0 && channels[i] >= 1 && channels[i] <= 65535 && (n_frames[i] == 0 || data_h[i]) &&
                        (uint64_t)n_frames[i] * 2 * (uint64_t)channels[i] <= 0xffffffffull - 36,
                    DCS_EINVAL, "dcs_wav_write_pcm16_async: file %d: bad frame / channel count (a wav holds < 4 GiB)", i);
    dcs_wav_batch* b = new dcs_wav_batch;
    b->remaining = n;
    std::vector<WavTask> tasks((size_t)n);
    for (int i = 0; i < n; ++i) {
        b->paths.emplace_back(paths[i]);
        status[i] = 1;
        WavTask& t = tasks[(size_t)i];
        t = WavTask{};
        t.batch = b;
        t.index = i;
        t.write = true;
        t.data = data_h[i];
        t.n_frames = n_frames[i];
        t.channels = channels[i];
        t.rate = rate[i];
        t.status = status + i;
    }
    p->push(tasks);
    *out = b;
    return DCS_OK;
}

DCS_API int dcs_wav_batch_done(dcs_wav_batch* b) {
    if (!b) return 1;
    std::lock_guard<std::mutex> lk(b->m);
    return b->remaining == 0 ? 1 : 0;
}

DCS_API int dcs_wav_batch_wait(dcs_wav_batch* b) {
    DCS_REQUIRE(b != nullptr, DCS_EINVAL, "dcs_wav_batch_wait: null batch");
    {
        std::unique_lock<std::mutex> lk(b->m);
        b->cv.wait(lk, [&] { return b->remaining == 0; });
    }
    delete b;
    return DCS_OK;
}

}  // extern "C"
