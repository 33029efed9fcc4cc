"""Print the engine's launch facts for the bench scene (run on the GPU box)."""
import sys; sys.path.insert(0, '.')
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
print(e.info())
