"""Ordered kernel list (name, workgroups, queue / stream, start, end) of the LAST forward in a rocprofv3 kernel trace of
scratch/rccl_timeline.py.  usage: rccl_timeline_summary.py TRACE_DIR OUT.json"""
import csv, glob, json, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'agg_tiled_flat4' in r['Kernel_Name']]
s = idx[-3] - 6                                   # the last forward: its three tile passes, the projections in front of them
t0 = int(rows[s]['Start_Timestamp'])
out = []
for r in rows[s:]:
    g, w = int(r['Grid_Size_X']), int(r['Workgroup_Size_X'])
    out.append({"start_us": round((int(r['Start_Timestamp']) - t0) / 1e3, 1), "end_us": round((int(r['End_Timestamp']) - t0) / 1e3, 1),
                "queue": r['Queue_Id'], "stream": r['Stream_Id'], "workgroups_x": g // w, "wg_size": w, "kernel": r['Kernel_Name'][:90]})
json.dump({"_how": "rocprofv3 --kernel-trace -- python scratch/rccl_timeline.py: rank 0's N = 8-size shard of cfg3 behind a ONE-rank nccl group "
           "with dist.FORCE_COLLECTIVES (the only communicator a 1-GPU lease allows), async logits concat; the last forward and the tail of "
           "the one before it.  At one rank RCCL turns the all-gather into a device copy on the communicator's stream and the in-place "
           "all-reduce into nothing - no channel kernel is launched; durations are inflated by the profiler.",
           "kernels": out}, open(sys.argv[2], 'w'), indent=0)
for k in out:
    print(k)
