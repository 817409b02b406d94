// Sanitizer driver for the host-side native code of libdcs (csrc/wavio.hip: the wav I/O thread pool of the batch-of-files driver).
// GPU AddressSanitizer is not available on this pool (xnack-), so the device code is covered by the guard-band harness
// (tests/test_gpu_guard.py); the HOST code that runs threads and touches caller memory is compiled here with g++ and
// -fsanitize=address,undefined / -fsanitize=thread (tests/test_sanitizers_cpu.py) and driven through the C ABI of include/dcs.h:
// batches of writes and reads in flight at once, odd and damaged files, buffers that are too small, missing directories, a batch
// collected late, the pool destroyed with work enqueued.  Exit code 0 = every result as expected (the sanitizers abort otherwise).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dcs.h"

static int fail(const char* what, long long a = 0, long long b = 0) {
    fprintf(stderr, "FAIL: %s (%lld, %lld)\n", what, a, b);
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 2) return fail("usage: wavio_san <scratch dir>");
    const std::string dir = argv[1];
    dcs_wav_pool* pool = nullptr;
    if (dcs_wav_pool_create(0, &pool) == DCS_OK) return fail("a pool of 0 threads was accepted");
    if (dcs_wav_pool_create(6, &pool) != DCS_OK || !pool) return fail("pool create");
    const int kFiles = 24, kRounds = 6;
    std::vector<std::vector<int16_t>> data(kFiles);
    std::vector<int64_t> frames(kFiles);
    std::vector<int32_t> chans(kFiles), rates(kFiles);
    unsigned seed = 1;
    for (int i = 0; i < kFiles; ++i) {
        chans[i] = 1 + i % 3;
        frames[i] = (i % 5 == 0) ? 0 : 1000 + 977 * i;
        rates[i] = i % 2 ? 22050 : 44100;
        data[i].resize((size_t)frames[i] * chans[i]);
        for (auto& v : data[i]) { seed = seed * 1664525u + 1013904223u; v = (int16_t)(seed >> 16); }
    }
    // several write batches in flight at once (into directories that do not exist yet), collected in reverse order
    std::vector<dcs_wav_batch*> batches;
    std::vector<std::vector<std::string>> paths(kRounds);
    std::vector<std::vector<int32_t>> status(kRounds, std::vector<int32_t>(kFiles, 12345));
    for (int r = 0; r < kRounds; ++r) {
        std::vector<const char*> cp;
        std::vector<const int16_t*> dp;
        for (int i = 0; i < kFiles; ++i) {
            paths[r].push_back(dir + "/r" + std::to_string(r) + "/d" + std::to_string(i % 4) + "/f" + std::to_string(i) + ".wav");
        }
        for (int i = 0; i < kFiles; ++i) { cp.push_back(paths[r][i].c_str()); dp.push_back(frames[i] ? data[i].data() : nullptr); }
        dcs_wav_batch* b = nullptr;
        if (dcs_wav_write_pcm16_async(pool, kFiles, cp.data(), dp.data(), frames.data(), chans.data(), rates.data(), status[r].data(), &b) != DCS_OK)
            return fail("write enqueue", r);
        batches.push_back(b);
    }
    for (int r = kRounds - 1; r >= 0; --r) {
        if (dcs_wav_batch_wait(batches[r]) != DCS_OK) return fail("write wait", r);
        for (int i = 0; i < kFiles; ++i)
            if (status[r][i] != 0) return fail("write status", r, status[r][i]);
    }
    // read everything back, all rounds in flight; one buffer per file exactly as large as needed, one a byte short
    std::vector<std::vector<std::vector<int16_t>>> got(kRounds, std::vector<std::vector<int16_t>>(kFiles));
    std::vector<std::vector<int32_t>> rrate(kRounds, std::vector<int32_t>(kFiles)), rch(kRounds, std::vector<int32_t>(kFiles)),
        rst(kRounds, std::vector<int32_t>(kFiles, 777));
    std::vector<std::vector<int64_t>> rfr(kRounds, std::vector<int64_t>(kFiles)), cap(kRounds, std::vector<int64_t>(kFiles));
    batches.clear();
    for (int r = 0; r < kRounds; ++r) {
        std::vector<const char*> cp;
        std::vector<void*> dst;
        for (int i = 0; i < kFiles; ++i) {
            got[r][i].assign(data[i].size() + 8, (int16_t)0x5A5A);                 // + a canary tail
            cap[r][i] = (int64_t)data[i].size() * 2 - ((r == 1 && i == 7) ? 1 : 0);   // (1, 7): too small by one byte
            cp.push_back(paths[r][i].c_str());
            dst.push_back(got[r][i].data());
        }
        dcs_wav_batch* b = nullptr;
        if (dcs_wav_read_pcm16_async(pool, kFiles, cp.data(), dst.data(), cap[r].data(), rrate[r].data(), rfr[r].data(), rch[r].data(),
                                     rst[r].data(), &b) != DCS_OK)
            return fail("read enqueue", r);
        batches.push_back(b);
    }
    for (int r = 0; r < kRounds; ++r) {
        while (!dcs_wav_batch_done(batches[r])) {}
        if (dcs_wav_batch_wait(batches[r]) != DCS_OK) return fail("read wait", r);
        for (int i = 0; i < kFiles; ++i) {
            if (r == 1 && i == 7) {
                if (rst[r][i] != 1) return fail("short buffer must be declined", rst[r][i]);
                continue;
            }
            if (rst[r][i] != 0 || rrate[r][i] != rates[i] || rfr[r][i] != frames[i] || (frames[i] && rch[r][i] != chans[i]))
                return fail("read result", r, i);
            if (!data[i].empty() && memcmp(got[r][i].data(), data[i].data(), data[i].size() * 2) != 0) return fail("read data", r, i);
            for (size_t k = data[i].size(); k < got[r][i].size(); ++k)
                if (got[r][i][k] != (int16_t)0x5A5A) return fail("write past the frames", r, i);
        }
    }
    // a missing file, a damaged header, a file cut inside its fmt chunk: verdicts, not crashes
    {
        const std::string junk = dir + "/junk.wav", cut = dir + "/cut.wav", missing = dir + "/missing.wav";
        FILE* fh = fopen(junk.c_str(), "wb"); fwrite("RIFF\0\0\0\0WAVEjunk", 1, 16, fh); fclose(fh);
        fh = fopen(cut.c_str(), "wb"); fwrite("RIFF\x64\0\0\0WAVEfmt \x10\0\0\0\x01\0\x02\0\x44", 1, 25, fh); fclose(fh);
        const char* cp[3] = {junk.c_str(), cut.c_str(), missing.c_str()};
        std::vector<int16_t> buf(64);
        void* dst[3] = {buf.data(), buf.data(), buf.data()};
        int64_t caps[3] = {128, 128, 128}, fr[3];
        int32_t rt[3], ch[3], st[3] = {9, 9, 9};
        dcs_wav_batch* b = nullptr;
        if (dcs_wav_read_pcm16_async(pool, 3, cp, dst, caps, rt, fr, ch, st, &b) != DCS_OK) return fail("odd enqueue");
        if (dcs_wav_batch_wait(b) != DCS_OK) return fail("odd wait");
        if (st[0] != 1 || st[1] != 1 || st[2] >= 0) return fail("odd verdicts", st[0] * 100 + st[1], st[2]);
    }
    // argument errors, and a pool destroyed with a batch enqueued and not yet collected (it must complete first and stay collectable)
    if (dcs_wav_write_pcm16_async(pool, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == DCS_OK) return fail("null arguments accepted");
    {
        std::vector<const char*> cp;
        std::vector<const int16_t*> dp;
        std::vector<std::string> ps;
        for (int i = 0; i < kFiles; ++i) ps.push_back(dir + "/late/f" + std::to_string(i) + ".wav");
        for (int i = 0; i < kFiles; ++i) { cp.push_back(ps[i].c_str()); dp.push_back(frames[i] ? data[i].data() : nullptr); }
        std::vector<int32_t> st(kFiles, 5);
        dcs_wav_batch* b = nullptr;
        if (dcs_wav_write_pcm16_async(pool, kFiles, cp.data(), dp.data(), frames.data(), chans.data(), rates.data(), st.data(), &b) != DCS_OK)
            return fail("late enqueue");
        dcs_wav_pool_destroy(pool);
        if (dcs_wav_batch_wait(b) != DCS_OK) return fail("late wait");
        for (int i = 0; i < kFiles; ++i)
            if (st[i] != 0) return fail("late status", i, st[i]);
    }
    printf("wavio sanitizer driver: %d write + %d read batches of %d files, odd files, late collection: ok\n", kRounds + 1, kRounds + 1, kFiles);
    return 0;
}
