"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; these helpers return its kernel dispatches and
counter values as the dict rows the CSV output used to give (the column names tools/summarize_trace.py and
tools/pmc_traffic.py read)."""
import csv
import glob
import gzip
import os
import sqlite3


def _db(path):
    if os.path.isdir(path):
        fs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
        return fs[0] if fs else None
    return path if path.endswith(".db") else None


def kernel_rows(path):
    db = _db(path)
    if db is None:
        return list(csv.DictReader(open(path)))
    c = sqlite3.connect(db)
    q = "select name, start, end, grid_x, workgroup_x, grid_y, dispatch_id, scratch_size, vgpr_count, lds_size from kernels order by start"
    return [{"Kernel_Name": r[0], "Start_Timestamp": r[1], "End_Timestamp": r[2], "Grid_Size_X": r[3], "Workgroup_Size_X": r[4],
             "Grid_Size_Y": r[5], "Dispatch_Id": r[6], "Scratch_Size": r[7], "VGPR_Count": r[8], "LDS_Size": r[9]} for r in c.execute(q)]


def counter_rows(path):
    db = _db(path)
    if db is None:
        fs = glob.glob(f"{path}/*counter_collection.csv*")
        if not fs:
            return []
        fh = gzip.open(fs[0], "rt") if fs[0].endswith(".gz") else open(fs[0])
        return list(csv.DictReader(fh))
    c = sqlite3.connect(db)
    q = "select dispatch_id, kernel_name, counter_name, value, grid_size, workgroup_size from counters_collection order by dispatch_id"
    return [{"Dispatch_Id": r[0], "Kernel_Name": r[1], "Counter_Name": r[2], "Counter_Value": r[3], "Grid_Size": r[4], "Workgroup_Size": r[5]}
            for r in c.execute(q)]


def short_name(kernel_name):
    """`void nbp_prep_kernel_spec<3>(int const*, ...)` -> `nbp_prep_kernel_spec<3>`"""
    n = kernel_name.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n
