#!/bin/bash
# renders in flight for small renders: 4 (the rule) against 6 and 8
cd "$GRAFT_REPO_ROOT"
for rep in 1; do for s in 4 6 8; do
  export JPT_PIPE_SLOTS=$s
  echo -n "slots=$s 1080p x1 "; python tools/rate.py 1920 1080 1 400 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
  echo -n "slots=$s C3/8 "; python tools/rate.py 1920 1080 8 400 8 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
  echo -n "slots=$s C3/4 "; python tools/rate.py 1920 1080 8 300 4 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
  echo -n "slots=$s C3/2 "; python tools/rate.py 1920 1080 8 200 2 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
  echo -n "slots=$s C2 "; python tools/rate.py 1280 720 4 300 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
  echo -n "slots=$s C3 "; python tools/rate.py 1920 1080 8 150 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
  echo -n "slots=$s 256x256x1 "; python tools/rate.py 256 256 1 500 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1
done; done
unset JPT_PIPE_SLOTS
python tools/enqueue_cost.py 2>&1 | grep enqueue
