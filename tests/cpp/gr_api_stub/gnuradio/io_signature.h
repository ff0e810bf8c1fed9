// COMPILE-CHECK STUB -- NOT GNU Radio (see pmt/pmt.h).
#pragma once
#include <memory>
#include <vector>

namespace gr {
class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    int min_streams, max_streams;
    std::vector<int> sizeof_stream_items;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        auto s = std::make_shared<io_signature>();
        s->min_streams = min_streams; s->max_streams = max_streams; s->sizeof_stream_items = {sizeof_stream_item};
        return s;
    }
    static sptr make2(int min_streams, int max_streams, int sizeof_stream_item1, int sizeof_stream_item2)
    {
        auto s = std::make_shared<io_signature>();
        s->min_streams = min_streams; s->max_streams = max_streams; s->sizeof_stream_items = {sizeof_stream_item1, sizeof_stream_item2};
        return s;
    }
};
}  // namespace gr
