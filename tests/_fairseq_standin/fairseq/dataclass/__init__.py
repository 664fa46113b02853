from dataclasses import dataclass
from typing import Optional


@dataclass
class FairseqDataclass:
    _name: Optional[str] = None
