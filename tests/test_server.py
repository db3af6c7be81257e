"""CPU: the minimal job server (README.md:42-81 of the reference; sequitr_amd/server.py): setup config
round trip, priority order, worker cap, GPU hand-out and the .job -> .job.complete rename, using the
GPU-free SERVER_test job."""
import json
import os

from sequitr_amd import core, server
from tests.test_jobs_config import write_job


def test_setup_writes_a_config_core_reads_back(tmp_path):
    fn = server.setup(str(tmp_path / "server.config"), jobdir="/jobs", logdir="/logs", outdir="/out", modeldir="/models")
    saved = {k: getattr(core.ServerConfiguration, k) for k in ("JOBDIR", "OUTDIR", "LOGDIR", "MAX_PROCESSES", "DEFAULT_GPUS")}
    try:
        assert core._configure(fn) == core.ServerConfiguration.VERSION
        assert core.ServerConfiguration.JOBDIR == "/jobs" and core.ServerConfiguration.OUTDIR == "/out"
        assert core.TensorflowConfiguration.MODELDIR == "/models"
        assert isinstance(core.ServerConfiguration.DEFAULT_GPUS, list) and core.ServerConfiguration.MAX_PROCESSES >= 1
    finally:
        for k, v in saved.items():
            setattr(core.ServerConfiguration, k, v)


def test_server_runs_jobs_by_priority_and_marks_them_complete(tmp_path):
    jobs, out = tmp_path / "jobs", tmp_path / "out"
    jobs.mkdir(), out.mkdir()
    for i, prio in enumerate((10, 99, 50)):
        write_job(jobs, func="SERVER_test", params=repr({"tag": i}), options="{}", ID="job%d" % i, priority=prio,
                  name="JOB_%d.job" % i)
    srv = server.Server(str(jobs), str(out), gpus=[0, 1], max_processes=2, delay=0.05)
    assert [j.ID for j in srv.pending()] == ["job1", "job2", "job0"]
    assert srv.poll_once() == 2                                            # worker cap
    assert sorted(g for _, g, _ in srv.running.values()) == [0, 1]         # one GPU each, least loaded first
    done = srv.serve(once=True)
    assert sorted(d[0] for d in done) == ["job0", "job1", "job2"] and all(rc == 0 for _, rc in done)
    assert sorted(os.listdir(str(jobs))) == ["JOB_%d.job.complete" % i for i in range(3)]
    for i in range(3):
        t = json.load(open(os.path.join(str(out), "JOB_job%d" % i, "test.json")))
        assert "tag" in t["params"]


def test_failed_and_unparsable_jobs_are_told_apart_from_finished_ones(tmp_path):
    """ADVICE r1 (server.reap): a job whose function raises (here: device = CPU, which this back end refuses loudly)
    ends as .job.failed with a non-zero worker exit code; a .job file that cannot be parsed becomes .job.invalid and is
    not polled (and logged) again every DELAY seconds; a good job still ends as .job.complete."""
    jobs, out = tmp_path / "jobs", tmp_path / "out"
    jobs.mkdir(), out.mkdir()
    write_job(jobs, func="SERVER_test", params="{}", options="{}", ID="good", name="JOB_good.job")
    write_job(jobs, func="SERVER_segment", params=repr({"input": {"synthetic": True}}), options="{}", ID="bad",
              device="CPU", name="JOB_bad.job")
    (jobs / "JOB_garbage.job").write_text("[job]\nID = x\n")               # header fields missing
    srv = server.Server(str(jobs), str(out), gpus=[0], max_processes=2, delay=0.05)
    done = dict(srv.serve(once=True))
    assert done["good"] == 0 and done["bad"] != 0
    assert sorted(os.listdir(str(jobs))) == ["JOB_bad.job.failed", "JOB_garbage.job.invalid", "JOB_good.job.complete"]
    assert srv.pending() == []


def test_a_job_file_caught_mid_write_is_not_discarded(tmp_path):
    """ADVICE r2 (server.pending): jobs are submitted by dropping files into the polled folder, so the first poll can
    see half a file.  It must be left alone (not renamed to .job.invalid -- the rename keeps the inode, the finished
    content would land under .invalid and never run), picked up once the writer finishes, and only a file that has
    stopped changing for `settle` seconds is invalidated."""
    import time
    jobs, out = tmp_path / "jobs", tmp_path / "out"
    jobs.mkdir(), out.mkdir()
    whole = write_job(jobs, func="SERVER_test", params="{}", options="{}", ID="late", name="JOB_tmp.job")
    text = open(whole).read()
    os.remove(whole)
    fn = jobs / "JOB_late.job"
    fn.write_text(text[:len(text) // 3])                                   # the copy has only started
    srv = server.Server(str(jobs), str(out), gpus=[0], max_processes=1, delay=0.05)
    srv.settle = 0.6
    for _ in range(3):                                                     # several polls inside the settle window
        assert srv.pending() == [] and os.path.exists(str(fn))
        time.sleep(0.05)
    fn.write_text(text)                                                    # the writer finishes
    assert [j.ID for j in srv.pending()] == ["late"] and srv.unparsed == {}
    done = dict(srv.serve(once=True))
    assert done == {"late": 0} and os.listdir(str(jobs)) == ["JOB_late.job.complete"]
    # a file that stays broken is invalidated, but only after it has been left alone for `settle` seconds
    bad = jobs / "JOB_bad.job"
    bad.write_text("[job]\nID = x\n")
    t0 = time.time()
    while os.path.exists(str(bad)):
        assert srv.pending() == []
        assert time.time() - t0 < 5
        time.sleep(0.05)
    assert time.time() - t0 >= 0.5 and os.path.exists(str(bad) + ".invalid")
