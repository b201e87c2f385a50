"""agent registry (counterpart of mani_skill/agents/registration.py)"""
from dataclasses import dataclass
from typing import Dict, List, Optional


@dataclass
class AgentSpec:
    agent_cls: type
    asset_download_ids: Optional[List[str]] = None


REGISTERED_AGENTS: Dict[str, AgentSpec] = {}


def register_agent(asset_download_ids: Optional[List[str]] = None, override=False):
    def deco(cls):
        if cls.uid in REGISTERED_AGENTS and not override:
            return cls
        REGISTERED_AGENTS[cls.uid] = AgentSpec(agent_cls=cls, asset_download_ids=asset_download_ids or [])
        return cls

    return deco
