// gz_source.h -- gzip input for the host-side readers: inflate off the parsing thread, and on many threads when
// the file allows it (at >= 40 Gbases/s on the GPU the inflate, not the kernel,
// bounds end-to-end seconds; one zlib stream inflates at ~0.35 GB/s on one core).
//
//   * BGZF (block gzip: bgzip, htslib, `samtools fastq | bgzip`; RFC 1952 members of <= 64 KB whose extra field
//     'BC' carries the member's size): the member boundaries are known without inflating, so batches of
//     members are inflated by a pool of threads and delivered in file order.
//   * any other gzip file (one deflate stream, or members of unknown size back to back): a single inflater
//     thread runs ahead of the parser -- inflate and parse overlap, the inflate itself stays serial (a deflate
//     stream has no entry points).
// Output is delivered as blocks of decompressed bytes, in order, through a bounded queue.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

class GzSource {
public:
    ~GzSource() { close(); }

    // true when `path` starts with the gzip magic; then the inflater threads are running
    bool open(const char *path, int threads)
    {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0 || st.st_size < 18) { ::close(fd_); fd_ = -1; return false; }
        size_ = (size_t)st.st_size;
        map_ = (const unsigned char *)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (map_ == (const unsigned char *)MAP_FAILED) { map_ = nullptr; ::close(fd_); fd_ = -1; return false; }
        if (!(map_[0] == 0x1f && map_[1] == 0x8b)) { close(); return false; }
        (void)madvise((void *)map_, size_, MADV_SEQUENTIAL);
        bgzf_ = member_size(0) > 0;
        n_threads_ = bgzf_ ? std::max(1, std::min(threads, 32)) : 1;
        if (bgzf_) {
            dispatcher_ = std::thread([this]() { dispatch_bgzf(); });
            for (int t = 0; t < n_threads_; t++) workers_.emplace_back([this]() { work_bgzf(); });
        } else {
            workers_.emplace_back([this]() { inflate_stream(); });
        }
        return true;
    }

    bool is_bgzf() const { return bgzf_; }
    int threads() const { return n_threads_; }

    // next block of decompressed bytes, in file order; false at the end (or after an error: ok() tells)
    bool next(std::vector<char> &out)
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_out_.wait(lk, [&] { return ready_.count(next_out_) || (producers_done_ && ready_.empty() && pending_ == 0) || failed_; });
        auto it = ready_.find(next_out_);
        if (it == ready_.end()) return false;
        out.swap(it->second);
        ready_.erase(it);
        next_out_++;
        lk.unlock();
        cv_room_.notify_all();
        return true;
    }
    bool ok() const { return !failed_; }

    void close()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_room_.notify_all(); cv_work_.notify_all(); cv_out_.notify_all();
        if (dispatcher_.joinable()) dispatcher_.join();
        for (auto &t : workers_) if (t.joinable()) t.join();
        workers_.clear();
        if (map_) munmap((void *)map_, size_);
        map_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }

private:
    struct Batch { int64_t seq; size_t lo, hi; };                // members [lo, hi) of the mapped file

    // size of the BGZF member at `at`, 0 when it is not one (RFC 1952 header with the 'BC' extra subfield)
    size_t member_size(size_t at) const
    {
        if (at + 18 > size_) return 0;
        const unsigned char *p = map_ + at;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
        const size_t xlen = p[10] | ((size_t)p[11] << 8);
        if (at + 12 + xlen > size_) return 0;
        for (size_t x = 0; x + 4 <= xlen;) {
            const unsigned char *f = p + 12 + x;
            const size_t slen = f[2] | ((size_t)f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) {
                const size_t bsize = (f[4] | ((size_t)f[5] << 8)) + 1;
                return (bsize >= 12 + xlen + 8 && at + bsize <= size_) ? bsize : 0;
            }
            x += 4 + slen;
        }
        return 0;
    }

    void fail()
    {
        { std::lock_guard<std::mutex> lk(mu_); failed_ = true; }
        cv_out_.notify_all(); cv_work_.notify_all(); cv_room_.notify_all();
    }

    // ---- BGZF: batches of members (~8 MB of output each) to the workers
    void dispatch_bgzf()
    {
        size_t at = 0;
        int64_t seq = 0;
        while (at < size_) {
            size_t lo = at, out_bytes = 0;
            while (at < size_ && out_bytes < ((size_t)8 << 20)) {
                const size_t m = member_size(at);
                if (!m) {
                    if (at == lo) { tail_at_ = at; goto done; }       // not a BGZF member: the rest is inflated as one stream
                    break;
                }
                const unsigned char *t = map_ + at + m - 4;
                out_bytes += t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                at += m;
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_room_.wait(lk, [&] { return stop_ || failed_ || (int64_t)(seq - next_out_) < 2 * n_threads_ + 2; });
                if (stop_ || failed_) break;
                work_.push_back(Batch{seq++, lo, at});
                pending_++;
            }
            cv_work_.notify_one();
        }
    done:
        if (tail_at_ != (size_t)-1 && !stop_ && !failed_) {
            // a trailing part that is not BGZF (rare: a plain member appended, or bytes that are no gzip at all): serial,
            // once every batch before it is inflated -- if the tail turns out to be no gzip member it is the end of the
            // input, as under gzread, and must not cost the batches still queued
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_out_.wait(lk, [&] { return stop_ || failed_ || pending_ == 0; });
            }
            if (!stop_ && !failed_) inflate_from(tail_at_, seq, seq > 0);
        }
        { std::lock_guard<std::mutex> lk(mu_); producers_done_ = true; }
        cv_work_.notify_all(); cv_out_.notify_all();
    }

    void work_bgzf()
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { fail(); return; }
        for (;;) {
            Batch b;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || failed_ || !work_.empty() || producers_done_; });
                if (stop_ || failed_ || work_.empty()) break;
                b = work_.front();
                work_.pop_front();
            }
            size_t total = 0;
            for (size_t at = b.lo; at < b.hi; at += member_size(at)) {
                const unsigned char *t = map_ + at + member_size(at) - 4;
                total += t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
            }
            std::vector<char> out(total);
            size_t o = 0;
            bool bad = false;
            for (size_t at = b.lo; at < b.hi && !bad;) {
                const size_t m = member_size(at);
                const unsigned char *p = map_ + at;
                const size_t xlen = p[10] | ((size_t)p[11] << 8);
                const size_t isize = p[m - 4] | ((size_t)p[m - 3] << 8) | ((size_t)p[m - 2] << 16) | ((size_t)p[m - 1] << 24);
                if (isize) {
                    inflateReset(&zs);
                    zs.next_in = const_cast<unsigned char *>(p + 12 + xlen);
                    zs.avail_in = (uInt)(m - 12 - xlen - 8);
                    zs.next_out = (unsigned char *)out.data() + o;
                    zs.avail_out = (uInt)isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.avail_out != 0) bad = true;
                    else {
                        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const unsigned char *)out.data() + o, (uInt)isize);
                        const uint32_t want = p[m - 8] | ((uint32_t)p[m - 7] << 8) | ((uint32_t)p[m - 6] << 16) | ((uint32_t)p[m - 5] << 24);
                        if (crc != want) bad = true;
                    }
                }
                o += isize;
                at += m;
            }
            if (bad) { inflateEnd(&zs); fail(); return; }
            {
                std::lock_guard<std::mutex> lk(mu_);
                ready_[b.seq].swap(out);
                pending_--;
            }
            cv_out_.notify_all();
        }
        inflateEnd(&zs);
    }

    // ---- one deflate stream (or members of unknown size): a single inflater ahead of the parser
    void inflate_stream()
    {
        inflate_from(0, 0);
        { std::lock_guard<std::mutex> lk(mu_); producers_done_ = true; }
        cv_out_.notify_all();
    }

    // after_members: whole members came before `at` -- bytes there that are no gzip member are then ignored, as gzread does
    void inflate_from(size_t at, int64_t seq, bool after_members = false)
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 16) != Z_OK) { fail(); return; }       // gzip wrapper
        zs.next_in = const_cast<unsigned char *>(map_ + at);
        size_t left = size_ - at;
        const size_t BLK = (size_t)4 << 20;
        int members = after_members ? 1 : 0;                              // members inflated to their end
        bool fresh = true;                                                // no byte of the current member's output yet
        for (bool end = false; !end;) {
            std::vector<char> out(BLK);
            size_t o = 0;
            while (o < BLK && !end) {
                if (zs.avail_in == 0) {
                    if (left == 0) { end = true; break; }
                    const size_t take = std::min<size_t>(left, (size_t)1 << 30);
                    zs.avail_in = (uInt)take;
                    left -= take;
                }
                zs.next_out = (unsigned char *)out.data() + o;
                zs.avail_out = (uInt)(BLK - o);
                const int rc = inflate(&zs, Z_NO_FLUSH);
                if (BLK - zs.avail_out > o) fresh = false;
                o = BLK - zs.avail_out;
                if (rc == Z_STREAM_END) {
                    // another member may follow (concatenated gzip): gzread goes on, so does this
                    members++;
                    fresh = true;
                    if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); fail(); return; }
                } else if (rc == Z_BUF_ERROR && zs.avail_in == 0 && left == 0) {
                    end = true;                                           // truncated file: what there is, as gzread
                } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                    if (members > 0 && fresh) { end = true; break; }      // bytes after the last member that are no gzip member: ignored, as gzread
                    inflateEnd(&zs); fail(); return;
                }
            }
            out.resize(o);
            if (o) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_room_.wait(lk, [&] { return stop_ || failed_ || (int64_t)(seq - next_out_) < 4; });
                if (stop_ || failed_) break;
                ready_[seq++].swap(out);
                lk.unlock();
                cv_out_.notify_all();
            }
        }
        inflateEnd(&zs);
    }

    int fd_ = -1;
    const unsigned char *map_ = nullptr;
    size_t size_ = 0;
    bool bgzf_ = false;
    int n_threads_ = 1;
    size_t tail_at_ = (size_t)-1;
    std::thread dispatcher_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_out_, cv_room_;
    std::deque<Batch> work_;
    std::map<int64_t, std::vector<char>> ready_;
    int64_t next_out_ = 0;
    int pending_ = 0;
    bool producers_done_ = false, stop_ = false;
    std::atomic<bool> failed_{false};
};
