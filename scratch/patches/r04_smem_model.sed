s|^#define WGNN_CLOB \(.*\)$|#define WGNN_CLOB \1, "s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79"|
s|^    const int wlane_addr = wstrip_addr + lane \* 4;|    const int wlane_addr = wstrip_addr + lane * 4; const int2* abl_ep = t.entries;|
s|\[mk\] "v"(row_mask), \[m\] "s"(m), \[sw\] "s"(sw)|[mk] "v"(row_mask), [m] "s"(m), [sw] "s"(sw), [ep] "s"(abl_ep)|
s|^        if (cs >= ce0) return;|        if (cs >= ce0) return; abl_ep = t.entries + max(ce0 - 64, 0);|
