"""A process that never touches the GPU and starts other programs on request.

pytest initialises the GPU when the session starts (tests/conftest.py); a program started from such a process would be a
fork + exec of a process that holds the device, which the GPU boxes refuse.  conftest.py therefore starts THIS helper
before anything touches the device, and multi-process GPU tests hand it their command lines: one JSON request per line on
stdin ({"commands": [[argv...], ...], "env": [{...}, ...], "cwd": str, "timeout": seconds}) -> the commands run
concurrently (a two-rank job is one request) -> one JSON reply per line on stdout ([{"returncode": int, "output": str}]).
"""
import json
import os
import subprocess
import sys
import tempfile


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        procs = []
        for argv, env in zip(req['commands'], req['env']):
            log = tempfile.TemporaryFile(mode='w+')
            procs.append((subprocess.Popen(argv, env={**os.environ, **env}, cwd=req.get('cwd'), stdout=log,
                                           stderr=subprocess.STDOUT), log))
        reply = []
        for proc, log in procs:
            try:
                rc = proc.wait(timeout=req.get('timeout', 600))
            except subprocess.TimeoutExpired:
                proc.kill()             # (the exact process this helper started)
                proc.wait()
                rc = -9
            log.seek(0)
            reply.append({'returncode': rc, 'output': log.read()[-20000:]})
            log.close()
        sys.stdout.write(json.dumps(reply) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
